"""Drop-in for the reference's `evaluation` package (scripts/test_body.py:16-17 imports `evaluation.FGD` and
`evaluation.metrics`): same names, the sums run on the GPU (talkshow_amd/evaluation.py, csrc/eval.hip)."""
