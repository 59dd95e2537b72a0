#!/bin/bash
# A/B builds of the attention translation unit for tools/probes/attn_ab.py (several libraries timed in ONE process):
#   build_attn_variants.sh NAME "GENERATOR FLAGS" [NAME "FLAGS" ...]     e.g.  a6 "--align 6"  a0 "--align 0"  p1 "--align 6 --pad 1"
#   NAME = base builds git HEAD's attention.hip + attention_r64_asm.inc instead (the committed kernel).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/self-forcing_amd/csrc"
make -s
mkdir -p "$ROOT/tools/probes/abl"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -fno-honor-nans -fno-honor-infinities"
OTHERS="gemm_bf16.o elementwise.o small_linear.o dit_forward.o conv_igemm.o conv_halo.o vae_elementwise.o vae_decode.o t5_encoder.o ceiling_probes.o capi.o"
while [ $# -ge 2 ]; do
  name=$1; gflags=$2; shift 2
  if [ "$name" = base ]; then
    git -C "$ROOT" show HEAD:self-forcing_amd/csrc/attention_r64_asm.inc > /tmp/r64_base.inc
    git -C "$ROOT" show HEAD:self-forcing_amd/csrc/attention.hip > /tmp/attention_base.hip
    cp /tmp/attention_base.hip ./_attention_base.hip
    /opt/rocm/bin/hipcc $FLAGS -DSF_R64_INC="\"/tmp/r64_base.inc\"" -c _attention_base.hip -o /tmp/att_base.o; rm -f _attention_base.hip
  else
    python "$ROOT/tools/gen_attention_r64.py" $gflags --out /tmp/r64_$name.inc > /dev/null
    /opt/rocm/bin/hipcc $FLAGS -DSF_R64_INC="\"/tmp/r64_$name.inc\"" -c attention.hip -o /tmp/att_$name.o
  fi
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/att_$name.o $OTHERS -o "$ROOT/tools/probes/abl/libattn_$name.so"
  echo built $name
done
