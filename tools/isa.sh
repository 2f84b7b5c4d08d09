#!/bin/bash
# tools/isa.sh <regex on mangled kernel name> : compile amr_pipeline.hip (K3 .. K5; pass another unit as $2) with -save-temps into /tmp/k1asm and
# print resource usage of matching kernels; the kernel ISA goes to /tmp/k1asm/<n>.s
set -e
mkdir -p /tmp/k1asm && cd /tmp/k1asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC ${EXTRA_FLAGS} -I/root/repo/include -I/root/repo/rtlamr_amd/csrc -save-temps -c /root/repo/rtlamr_amd/csrc/${2:-amr_pipeline.hip} -o /tmp/k1asm/a.o
S=amrdemod-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - "$1" <<'PY'
import re,sys
pat=re.compile(sys.argv[1])
lines=open('/tmp/k1asm/amrdemod-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
i=0
names=[]
for n,l in enumerate(lines):
    m=re.match(r'^(_Z\w+):',l)
    if m and pat.search(m.group(1)):
        name=m.group(1)
        end=next(k for k in range(n,len(lines)) if lines[k].startswith('.Lfunc_end'))
        body=lines[n:end]
        meta={}
        for k in range(end,min(end+120,len(lines))):
            for key in ('next_free_vgpr','next_free_sgpr','group_segment_fixed_size','private_segment_fixed_size'):
                mm=re.search(r'\.amdhsa_'+key+r'\s+(\d+)',lines[k])
                if mm and key not in meta: meta[key]=mm.group(1)
        fn=f'/tmp/k1asm/k{len(names)}.s'
        open(fn,'w').write('\n'.join(body))
        cnt=lambda p: sum(1 for b in body if re.search(p,b))
        print(name[:60],meta,'lines',len(body),'v_mov',cnt(r'\bv_mov_b'),'waitcnt',cnt('s_waitcnt'),'ds_read_b32',cnt('ds_read_b32'),'scratch',cnt('scratch_'),'->',fn)
        names.append(name)
PY
