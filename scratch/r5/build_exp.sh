#!/bin/sh
# builds libwino_exp<k>.so for the bound-splitting experiments of wino_exp.hip (scratch copy of csrc/conv_wino.hip)
cd "$(dirname "$0")"
for k in ${EXPS:-0 1 2 3 4 5}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I../../include -I../../dynmm_amd/csrc -Wno-inline-asm -DPIPE=$k wino_pipe.hip -o libwino_exp$k.so &
done
wait
ls -la libwino_exp*.so
