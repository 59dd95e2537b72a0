#!/bin/bash
# Build timing-only variants of the C-ABI library with compile-time ablation macros in attention.hip
# (usage: build_abl.sh NAME "-DSF_ABL_X ..." ...).  The variant is selected at run time by SF_HIP_LIB.
set -e
cd "$(dirname "$0")/../../self-forcing_amd/csrc"
make -s
mkdir -p ../../tools/probes/abl
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize \
     -fno-honor-nans -fno-honor-infinities $flags -c attention.hip -o /tmp/abl_att_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gemm_bf16.o /tmp/abl_att_$name.o elementwise.o \
     small_linear.o dit_forward.o conv_igemm.o vae_elementwise.o vae_decode.o t5_encoder.o capi.o -o ../../tools/probes/abl/libabl_$name.so
  echo built $name
done
