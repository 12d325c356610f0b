#!/bin/bash
# kernel resource + instruction-mix report:  tools/kstat.sh [kernel-substring] [extra hipcc flags...]
k=${1:-wbc_hex_kernelILi1E}; shift
mkdir -p /tmp/asm; cd /tmp/asm
/opt/rocm/bin/hipcc $(cat /root/repo/quadruped_drake_amd/csrc/hipcc_flags.txt) "$@" -S --cuda-device-only -o ks.s /root/repo/quadruped_drake_amd/csrc/wbc_kernels.hip 2>/dev/null || exit 1
python3 - "$k" <<'PY'
import sys,re,collections
k=sys.argv[1]
txt=open('/tmp/asm/ks.s').read()
# metadata
for m in re.finditer(r'\.name:\s+(\S+)\n((?:\s+\..*\n)+)', txt):
    pass
md=re.findall(r'- \.agpr_count:.*?\.wavefront_size', txt, re.S)
for blk in md:
    nm=re.search(r'\.name:\s+(\S+)',blk).group(1)
    if k in nm:
        g=lambda f: re.search(r'\.%s:\s+(\d+)'%f,blk).group(1)
        print(nm[:40],'vgpr',g('vgpr_count'),'agpr',g('agpr_count'),'sgpr',g('sgpr_count'),'scratch',g('private_segment_fixed_size'),'lds',g('group_segment_fixed_size'))
h=collections.Counter(); on=False
for line in txt.split('\n'):
    if line.startswith('_Z') and k in line.split(':')[0] and ':' in line: on=True; continue
    if on and line.startswith('.Lfunc_end'): break
    if on:
        s=line.strip()
        if not s or s[0] in ';.' or s.endswith(':'): continue
        op=s.split()[0]
        if 'dpp' in s and not op.endswith('dpp'): op+='_dpp'
        h[op]+=1
tot=sum(h.values())
grp=collections.Counter()
for op,c in h.items():
    if re.match(r'v_(fma|fmac|mul|add)_f64',op): grp['f64 math']+=c
    elif 'dpp' in op: grp['dpp']+=c
    elif 'cndmask' in op: grp['cndmask']+=c
    elif 'accvgpr' in op: grp['agpr moves']+=c
    elif op.startswith('scratch'): grp['scratch']+=c
    elif op.startswith('ds_'): grp['lds']+=c
    elif op.startswith('global') or op.startswith('flat'): grp['global']+=c
    elif op.startswith('s_'): grp['scalar']+=c
    elif op.startswith('v_mov'): grp['v_mov']+=c
    else: grp['other valu']+=c
print('total',tot, dict(grp.most_common()))
print(' '.join('%s:%d'%(o,c) for o,c in h.most_common(28)))
PY
