#!/bin/bash
# Per-section (; MARK) instruction / scratch / barrier counts of the group kernels (N = 16384, 32768) from a -save-temps build.
set -e
B=/root/repo/amcpy_amd/csrc/build
mkdir -p $B && cd $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-math-errno -fno-slp-vectorize $EXTRA \
  -save-temps ../amcx.hip -o $B/libamcx_tmp.so 2> $B/remarks.txt || { grep -E "error" -A5 $B/remarks.txt; exit 1; }
for W in 8 16; do
  k=$(grep -o "_ZN4amcx5group28amcx_features18_group_kernelILi${W}E[A-Za-z0-9_]*" amcx-hip-amdgcn-amd-amdhsa-gfx950.s | sort -u | head -1)
  awk -v k="$k:" '$1==k{p=1} p{print} p&&/s_endpgm/{exit}' amcx-hip-amdgcn-amd-amdhsa-gfx950.s > group_$W.s
  echo "== W=$W ($(wc -l < group_$W.s) lines; $(grep -E "vgpr_count|vgpr_spill|private_segment_fixed" amcx-hip-amdgcn-amd-amdhsa-gfx950.s | grep -A0 -B0 . | head -0))"
  awk '/MARK/{sec=$3; n++; key=n":"sec; order[n]=key} /scratch_load/{l[key]++} /scratch_store/{s[key]++} /^[ \t]+v_/{v[key]++} /^[ \t]+ds_/{d[key]++} /s_barrier/{b[key]++}
       END{for(i=1;i<=n;i++){k=order[i]; printf "%-12s valu %5d ds %4d sload %4d sstore %4d barriers %d\n", k, v[k], d[k], l[k], s[k], b[k]}}' group_$W.s
done
