"""Analysis: list the loops (backward branches) of one kernel in an assembly listing and their instruction mix."""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
txt = open(path).read().split('\n')
start = [i for i, l in enumerate(txt) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l][0]
end = [i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end')][0]
labels = {}; ins = []
for l in txt[start + 1:end]:
    s = l.split(';')[0].strip()
    if not s: continue
    if s.endswith(':'):
        labels[s[:-1]] = len(ins); continue
    if s.startswith('.'): continue
    ins.append(s)
print("instructions", len(ins))
def mix(seg):
    h = collections.Counter()
    for s in seg:
        op = s.split()[0]
        if re.match(r'v_(fma|fmac|mul|add)_f64', op): h['f64'] += 1
        elif 'dpp' in s: h['dpp'] += 1
        elif 'cndmask' in op: h['cndmask'] += 1
        elif 'accvgpr' in op: h['agpr'] += 1
        elif op.startswith('ds_'): h['lds'] += 1
        elif op.startswith('s_'): h['scalar'] += 1
        elif op.startswith('v_mov'): h['v_mov'] += 1
        else: h['valu_other'] += 1
    return dict(h.most_common())
for i, s in enumerate(ins):
    m = re.match(r's_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)', s)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            print('loop', labels[t], '->', i, 'len', i - labels[t], mix(ins[labels[t]:i]))
if len(sys.argv) > 4:
    lo, hi = int(sys.argv[3]), int(sys.argv[4])
    inv = collections.defaultdict(list)
    for k, v in labels.items(): inv[v].append(k)
    blk = lo
    for i in range(lo, hi + 1):
        s = ins[i]
        if i in inv and i > blk:
            print("  [%d..%d) %d instr %s" % (blk, i, i - blk, mix(ins[blk:i]))); blk = i
        if i in inv: print("%s:" % ",".join(inv[i]))
        if s.startswith('s_cbranch') or s.startswith('s_branch'):
            print("  [%d..%d] %d instr %s" % (blk, i, i - blk + 1, mix(ins[blk:i + 1]))); print("     ", s); blk = i + 1
