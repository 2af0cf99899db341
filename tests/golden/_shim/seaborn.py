"""TEST INFRASTRUCTURE ONLY: the reference script imports seaborn and never calls it."""


def histplot(*args, **kwargs):  # (badsunks_AR.py:40, :80 draw a histogram that is never saved: nothing of the output depends on it)
    return None
