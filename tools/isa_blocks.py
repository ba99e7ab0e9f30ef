"""Instruction mix per basic block of one kernel in a hipcc -S listing.  usage: isa_blocks.py <file.s> <substring of the kernel symbol> [min]"""
import collections, re, sys
L = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(L) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l][0]
end = [i for i in range(start, len(L)) if 's_endpgm' in L[i]][0]
def cls(op):
    if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')): return 'trans'
    if op.startswith('v_pk_'): return 'pk'
    if op.startswith('v_mfma') or op.startswith('v_smfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    return 'other'
seg = [['entry', collections.Counter()]]
for l in L[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';'): continue
    if re.match(r'^\.LBB\d+_\d+:', t):
        seg.append([t[:70], collections.Counter()]); continue
    if t.startswith('.'): continue
    seg[-1][1][cls(t.split()[0])] += 1
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = collections.Counter()
for n, cc in seg:
    tot.update(cc)
    if sum(cc.values()) >= lo: print(n, sum(cc.values()), dict(cc))
print("total", sum(tot.values()), dict(tot))
