#!/bin/bash
# Timing-only variants of the hand-scheduled attention kernel: build_r64_abl.sh NAME ABL[,ABL...] ...
# (ablations of tools/gen_attention_r64.py: nosoftmax nolds nomfma nodma nobarrier).  Select with SF_HIP_LIB.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/self-forcing_amd/csrc"
make -s
mkdir -p "$ROOT/tools/probes/abl"
while [ $# -ge 2 ]; do
  name=$1; abl=$2; shift 2
  python "$ROOT/tools/gen_attention_r64.py" --abl "$abl" --out /tmp/r64_$name.inc > /dev/null
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize \
     -fno-honor-nans -fno-honor-infinities -DSF_R64_INC="\"/tmp/r64_$name.inc\"" -c attention.hip -o /tmp/abl_att_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm_bf16.o /tmp/abl_att_$name.o elementwise.o \
     small_linear.o dit_forward.o conv_igemm.o conv_halo.o vae_elementwise.o vae_decode.o t5_encoder.o capi.o -o "$ROOT/tools/probes/abl/libabl_$name.so"
  echo built $name
done
