"""Names the reference's `nets` package exports that the hot-path scope leaves out (the Habibie et al.
baseline).  Constructing them says so instead of failing with an ImportError."""


class _OutOfScope:
    _what = ""

    def __init__(self, args=None, config=None, *a, **k):
        raise NotImplementedError(
            f"{self._what} is outside the speech->SMPL-X inference hot path this package implements "
            "(SURVEY.md §2/§8: evaluation / baseline component); use the reference implementation for it.")


class LS3DCG(_OutOfScope):
    _what = "LS3DCG (Habibie et al. baseline, nets/LS3DCG.py)"
