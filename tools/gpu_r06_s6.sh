#!/bin/bash
# r06 session 6: stream-K band — single-layer A/B, then the face pass per layer with TS_CONV_SK=0 / 1
mkdir -p gpurun_out/r06_s6
O=gpurun_out/r06_s6
timeout 600 python tools/sk_layers.py > $O/sk_layers.txt 2>$O/sk_layers.err
cat $O/sk_layers.txt; tail -3 $O/sk_layers.err
for sk in 0 1; do
  TS_CONV_SK=$sk timeout 300 python tools/face_layers.py 2>$O/face_layers_sk$sk.err | tail -1 > $O/face_sk$sk.txt
  cat $O/face_sk$sk.txt
done
