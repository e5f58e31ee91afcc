#!/bin/bash
# Static instruction counts of the NTT block kernels: VALU instructions, v_mad_u64_u32 and v_mov_b32 per kernel
# and v_mad_u64_u32 per barrier interval.  A pseudo-Mersenne butterfly is 8 multiply-adds (7 in the product and
# its folds, 1 in the fold of x), so a register pass of 4 stages has 4 * 8 * 8 = 256 per thread: anything else
# means the compiler lost a 32-bit fact somewhere (this is how the doubled multiplications of the fused-product
# kernel were found).   usage: tools/count_mads.sh [file.hip] [symbol-regex]
set -e
cd "$(dirname "$0")/../primus-fhe_amd"
src=${1:-csrc/pfhe_ntt.hip}; pat=${2:-ntt_block_kernelINS_7PmArithELi12}
out=$(mktemp /tmp/pfhe_isa_XXXXXX.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -S --cuda-device-only -o $out $src 2>/dev/null
grep -n "^_ZN4pfhe.*${pat}.*:" $out | while IFS=: read L rest; do
  E=$(awk -v s=$L 'NR>=s && /s_endpgm/ {print NR; exit}' $out)
  name=$(echo "$rest" | sed 's/ *;.*//' | c++filt | sed 's/pfhe::(anonymous namespace):://; s/pfhe:://g; s/(.*//' | cut -c1-70)
  sed -n "${L},${E}p" $out > $out.k
  printf "%-72s valu %5d  mad %5d  mov %5d  per barrier interval:" "$name" "$(grep -cE '^\s+v_' $out.k)" "$(grep -c v_mad_u64_u32 $out.k)" "$(grep -c v_mov_b32 $out.k)"
  awk '/s_barrier/{n++} {c[n]+= ($1=="v_mad_u64_u32")} END{for(i=0;i<=n;i++) printf " %d", c[i]; print ""}' $out.k
done
rm -f $out $out.k
