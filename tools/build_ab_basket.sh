#!/bin/bash
# Builds the basket A/B variants timed by tools/ab_basket.py (one .so each).
set -e
cd "$(dirname "$0")/../montecarlocuda_amd/csrc"
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -w -shared"
/opt/rocm/bin/hipcc $F -DMC_AB_BASKET_SGPR -o ../../tools/abk_0_sgpr.so mc_api.hip &
/opt/rocm/bin/hipcc $F -DMC_AB_FENCE_PERIOD=0 -o ../../tools/abk_1_lds_nofence.so mc_api.hip &
/opt/rocm/bin/hipcc $F -DMC_AB_FENCE_PERIOD=1 -o ../../tools/abk_2_lds_fence1.so mc_api.hip &
/opt/rocm/bin/hipcc $F -DMC_AB_FENCE_PERIOD=2 -o ../../tools/abk_3_lds_fence2.so mc_api.hip &
wait
/opt/rocm/bin/hipcc $F -DMC_AB_FENCE_PERIOD=4 -o ../../tools/abk_4_lds_fence4.so mc_api.hip &
/opt/rocm/bin/hipcc $F -o ../../tools/abk_5_default.so mc_api.hip &
wait
ls -la ../../tools/abk_*.so
