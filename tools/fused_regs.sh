#!/bin/bash
# register / scratch report of the one-launch layer kernel per body variant (each compiled alone)
cd /root/repo/geomae_amd/csrc
for n in "$@"; do python - $n <<'P' > _tmp_fx.hip
import sys
n=int(sys.argv[1])
s=open('sst_fused.hip').read()
for k in (1,2,3,4):
    if k!=n: s=s.replace('case %d: fused_fwd_body<%d, true>(A, s0, T, nt, lds); break;'%(k,k),'')
if n!=9: s=s.replace("default: fused_fwd_body<9, false>(A, s0, T, nt, lds); break;","default: break;")
print(s)
P
echo "only <$n>:"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -c _tmp_fx.hip -o /tmp/fx/x.o 2>&1 | grep -E " VGPRs:|Scratch|VGPRs Spill"; done; rm -f _tmp_fx.hip
