#!/bin/bash
# RETIRED EXPERIMENT (round 1), kept for reference: timing-only builds of the warp-specialised attention kernel.
#   build_ws_abl.sh NAME ABL[,ABL...] ...   (ablations of gen_attention_ws.py: nosoftmax nodma nobarrier nopv noqk; "" = full)
# The shipped library does not contain this kernel.  The translation unit built here #includes the product's
# attention.hip (same layouts / helpers) followed by attention_ws_kernel.inc and exports the probe-only symbol
# sf_attention_ws (same arguments as sf_attention); load tools/probes/abl/libabl_NAME.so with ctypes to time it.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../../.." && pwd)"
cd "$ROOT/self-forcing_amd/csrc"
make -s
mkdir -p "$ROOT/tools/probes/abl"
while [ $# -ge 2 ]; do
  name=$1; abl=$2; shift 2
  python "$HERE/gen_attention_ws.py" --abl "$abl" --out /tmp/ws_$name.inc > /dev/null
  printf '#include "%s"\n#include "%s"\n#include "%s"\n' "/tmp/ws_$name.inc" "$ROOT/self-forcing_amd/csrc/attention.hip" \
     "$HERE/attention_ws_kernel.inc" > /tmp/ws_tu_$name.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize \
     -fno-honor-nans -fno-honor-infinities -I"$ROOT/self-forcing_amd/csrc" -c /tmp/ws_tu_$name.hip -o /tmp/abl_att_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm_bf16.o /tmp/abl_att_$name.o elementwise.o \
     small_linear.o dit_forward.o conv_igemm.o vae_elementwise.o vae_decode.o t5_encoder.o capi.o -o "$ROOT/tools/probes/abl/libabl_$name.so"
  echo built $name
done
