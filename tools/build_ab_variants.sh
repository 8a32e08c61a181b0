#!/bin/bash
# Builds the A/B variants of the engine that tools/ab_f64.py times in one process (one .so each).
set -e
cd "$(dirname "$0")/../montecarlocuda_amd/csrc"
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -w -shared"
/opt/rocm/bin/hipcc $F -o ../../tools/ab_0_default.so mc_api.hip
/opt/rocm/bin/hipcc $F -DMC_AB_OCML_EXP -o ../../tools/ab_1_ocml_exp.so mc_api.hip
/opt/rocm/bin/hipcc $F -DMC_AB_CONST_COEF -o ../../tools/ab_2_const_coef.so mc_api.hip
/opt/rocm/bin/hipcc $F -DMC_AB_OCML_EXP -DMC_AB_CONST_COEF -o ../../tools/ab_3_ocml_const.so mc_api.hip
ls -la ../../tools/ab_*.so
