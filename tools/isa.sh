#!/bin/bash
# Build amcx.hip with -save-temps and summarise the wave kernel's ISA per MARK section.
set -e
B=/root/repo/amcpy_amd/csrc/build
mkdir -p $B && cd $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-math-errno -fno-slp-vectorize $EXTRA \
  -save-temps -Rpass-analysis=kernel-resource-usage ../amcx.hip -o $B/libamcx_tmp.so 2> $B/remarks.txt || { grep -E "error" -A5 $B/remarks.txt; exit 1; }
grep -A14 "wave_kernelILi" $B/remarks.txt | grep -E "Function Name|VGPRs|Scratch|Occupancy|SGPRs:" | sed 's/\[-Rpass.*//'
for k in $(grep -o "_ZN4amcx4wave27amcx_features18_wave_kernelILi[0-9]*E[A-Za-z0-9_]*" amcx-hip-amdgcn-amd-amdhsa-gfx950.s | sort -u); do
  awk -v k="$k:" '$1==k{p=1} p{print} p&&/s_endpgm/{exit}' amcx-hip-amdgcn-amd-amdhsa-gfx950.s > $B/wave_$(echo $k | grep -o "ILi[0-9]*" | tr -d ILi).s
done
for f in $B/wave_*.s; do
  echo "== $f ($(wc -l < $f) lines)"
  awk '/MARK/{sec=$3; order[++n]=sec} /scratch_/{c[sec]++} /^[ \t]+v_/{v[sec]++} /^[ \t]+ds_/{d[sec]++} /^[ \t]+s_nop/{sn[sec]++} /^[ \t]+s_waitcnt/{w[sec]++}
       END{for(i=1;i<=n;i++){k=order[i]; if(!(k in seen)){seen[k]=1; printf "%-10s valu=%-5d ds=%-4d scratch=%-4d s_nop=%-4d waitcnt=%-4d\n", k, v[k], d[k], c[k], sn[k], w[k]}}}' $f
done
