"""TEST INFRASTRUCTURE ONLY: the reference script imports seaborn and never calls it."""
