"""Placeholder for `nets/smplx_face.py` (wav2vec2-based face generator).

The face path is the next kernel family to land (DESIGN.md "what comes next"); until then constructing the wrapper
raises with that explanation rather than silently running something else.
"""
from nets.base import TrainWrapperBaseClass


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        raise NotImplementedError(
            "s2g_face (wav2vec2 encoder + LayerNorm conv heads) is not built yet in the MI355X-native path: "
            "SURVEY.md §8 rows a11-a13 are scheduled after the body path (DESIGN.md §6).")
