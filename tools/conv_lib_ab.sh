#!/bin/bash
# same-box A/B of library builds on the conv stacks: conv_lib_ab.sh head band sb  (tools/lib_<name>.so), two rounds each
cp talkshow_amd/lib/libtalkshow_hip.so /tmp/lib_keep.so
for round in 1 2; do
for v in "$@"; do
  cp tools/lib_$v.so talkshow_amd/lib/libtalkshow_hip.so
  bash tools/conv_mix_ab.sh "TS_LIB=$v" | head -1
  [ -n "$TS_TUNE" ] && TS_TUNE_FEW=1 TS_TILES=${TS_TILES:-1} TS_B=${TS_TUNE_B:-32} python tools/tune_conv.py 2>&1 | grep -E "4096|1024->1024"
done; done
cp /tmp/lib_keep.so talkshow_amd/lib/libtalkshow_hip.so
