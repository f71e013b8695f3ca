#!/usr/bin/env python3
"""Copies the reference's own demo recordings — format fixtures it ships under demo_audio/ — into tests/golden/audio/ (git-ignored:
they are the reference's files, not this repository's; `__graft_entry__.build()` runs this where /root/reference exists and the
copies travel to the GPU box with the snapshot, like oracle/_ref) so that the GPU box, which has no /root/reference, can run the
wav-in path on real speech: style.wav (22 kHz stereo int16, exactly 10.0 s),
1st-page.wav (16 kHz mono, 12.816 s), french.wav (24 kHz mono, 9.6125 s).  Writes audio_manifest.json with their sha256, sample
rate, shape and the frame counts SURVEY.md §8(c) documents for them (demo/*/*.npy hold 300 / 384 / 288 rows of 265 values).

    python tests/golden/audio/fetch_reference_audio.py [/root/reference]
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
from scipy.io import wavfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from talkshow_amd import frontend as fe  # noqa: E402   (host twin of librosa.load(sr=16000): the face input of the real-audio goldens)

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
FRAMES = {"style.wav": 300, "1st-page.wav": 384, "french.wav": 288}

manifest = {}
for name, frames in FRAMES.items():
    src = os.path.join(REF, "demo_audio", name)
    dst = os.path.join(HERE, name)
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)
    sr, a = wavfile.read(dst)
    # the 16 kHz mono samples the face path reads (tests/golden/real_audio_face.npz stores their sha256): native for 1st-page.wav,
    # the host twin of kaiser_best for the other two — written next to the recording, git-ignored like it
    wav16 = fe.get_wav16(dst, host=True)[:, 0]
    np.save(dst + ".wav16.npy", wav16)
    manifest[name] = {"sha256": hashlib.sha256(open(dst, "rb").read()).hexdigest(), "sample_rate": int(sr), "shape": list(a.shape),
                      "dtype": str(a.dtype), "seconds": a.shape[0] / sr, "frames_30fps": frames,
                      "wav16_samples": int(wav16.shape[0]), "wav16_sha256": hashlib.sha256(wav16.tobytes()).hexdigest(),
                      "source": "yhw-yhw/TalkSHOW demo_audio/" + name}
json.dump(manifest, open(os.path.join(HERE, "audio_manifest.json"), "w"), indent=1)
print(json.dumps(manifest, indent=1))
