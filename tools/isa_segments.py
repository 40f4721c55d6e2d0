"""Instruction mix of a kernel's ISA between consecutive s_barrier instructions (straight-line bodies).
usage: python tools/isa_segments.py file.s kernel_name_substring"""
import collections, sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and name in l and ':' in l][0]
end = [i for i, l in enumerate(lines) if i > start and 's_endpgm' in l][0]
seg = 0
cnt = collections.defaultdict(collections.Counter)
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.split(';')[0].strip().endswith(':'):
        continue
    op = t.split()[0]
    if op == 's_barrier':
        seg += 1
        continue
    if op.startswith('v_mfma'): cls = 'mfma'
    elif op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_log', 'v_sqrt')): cls = 'trans'
    elif op.startswith('v_'): cls = 'valu'
    elif op.startswith('s_waitcnt'): cls = 'wait'
    elif op.startswith('s_'): cls = 'salu'
    elif op.startswith('ds_'): cls = 'lds'
    elif op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): cls = 'vmem'
    else: cls = 'other'
    cnt[seg][cls] += 1
for k in sorted(cnt):
    print(k, dict(cnt[k]), sum(cnt[k].values()))
