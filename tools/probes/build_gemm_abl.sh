#!/bin/bash
# Timing-only variants of gemm_bf16.hip: build_gemm_abl.sh NAME "-DSF_ABL_X ..." ...  (select with SF_HIP_LIB)
set -e
cd "$(dirname "$0")/../../self-forcing_amd/csrc"
make -s
mkdir -p ../../tools/probes/abl
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $flags -c gemm_bf16.hip -o /tmp/abl_gemm_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/abl_gemm_$name.o attention.o elementwise.o \
     small_linear.o dit_forward.o conv_igemm.o vae_elementwise.o vae_decode.o t5_encoder.o capi.o -o ../../tools/probes/abl/libabl_$name.so
  echo built $name
done
