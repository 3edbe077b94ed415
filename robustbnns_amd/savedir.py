"""Where the drivers write their side-effect files.

The reference's drivers put pickles / CSVs / PNGs under three relative directories (reference savedir.py:4-6) and the
callers of this package rely on the same three names: posteriors and cached loss gradients under DATA, figures under
PLOTS, and attack results under a TESTS directory that carries the date of the run (fixed when this module is imported,
so one run writes to one directory even across midnight).
"""
import datetime as _datetime

_RUN_DATE = _datetime.date.today().isoformat()          # YYYY-MM-DD, same stamp as time.strftime('%Y-%m-%d')


def _under(*parts):
    return "/".join(parts) + "/"


DATA, PLOTS, TESTS = _under("data"), _under("plots"), _under("tests", _RUN_DATE)
