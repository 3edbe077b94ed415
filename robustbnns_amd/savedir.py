"""Output locations, same names as the reference's savedir.py:4-6 (TESTS is date-stamped at import)."""
import time

DATA = "data/"
PLOTS = "plots/"
TESTS = "tests/" + str(time.strftime('%Y-%m-%d')) + "/"
