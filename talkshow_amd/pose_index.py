"""Index bookkeeping between the 165-d SMPL-X axis-angle vector and the 129 modelled dims.

Restates `data_utils/lower_body.py:44-56` (`c_index_3d`): of the 165 pose dims, the jaw/eyes (0-8), global
orientation + lower body joints (9-17, 21-26, 30-35) and dims 45-50 are held fixed; the remaining 129 are what the
body (first 39) and hand (last 90) VQ-VAEs model.  The reference's off-by-six between this list and `part2full`
 is caller-side behaviour and is deliberately not "fixed" here.
"""
import numpy as np

FIX_INDEX_3D = list(range(0, 18)) + list(range(21, 27)) + list(range(30, 36)) + list(range(45, 51))
_all = np.ones(165)
_all[FIX_INDEX_3D] = 0
c_index_3d = np.asarray([i for i, v in enumerate(_all) if v == 1])
assert c_index_3d.shape == (129,)
