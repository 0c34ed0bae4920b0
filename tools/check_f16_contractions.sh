#!/usr/bin/env bash
# Lists every kernel whose gfx950 code holds a v_fma_mixlo_f16 / v_fma_mixhi_f16 (a product-sum rounded ONCE to fp16: hipcc's default
# -ffp-contract turns `(f16)fmaf(a, b, c)` into it) or a v_cvt_pkrtz_f16_f32 (round toward zero).  Two kernels documented as bit-identical
# must agree on where they round: round 5 found the fused q-projection + cross-attention launch rounding q once where the GEMM route
# rounds twice (fp32, then fp16) -- 1 element in ~15,000, fp16 only.  Expected output today: the three LSTM kernels
# (h = o * tanh(c), fused alike in all three) and gemm_f32x (a multiplication by 2^11: exact either way).  Runs in the build container
# (hipcc cross-compiles; ~10 minutes for all sources):   bash tools/check_f16_contractions.sh
set -uo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R/deephumor_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-comment --cuda-device-only -S -I. -I"$R/include" "$f" -o /tmp/_dh_$f.s 2>/dev/null || { echo "$f: did not compile"; continue; }
  awk -v src="$f" '/^[_A-Za-z0-9]+:/{name=$1} /v_fma_mixlo_f16|v_fma_mixhi_f16|v_cvt_pkrtz_f16_f32/{c[name]++} END{for(n in c) print src, c[n], n}' /tmp/_dh_$f.s
  rm -f /tmp/_dh_$f.s
done
