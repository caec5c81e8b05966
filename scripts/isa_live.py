"""Rough VGPR liveness profile of one kernel of a gfx950 listing (straight-line approximation: a register is live from a
write to its last read before the next write): python isa_live.py f.s NAME  -> live count every N lines + markers"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^\S*%s\S*:.*$' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():]
body = body[:re.search(r'^\.Lfunc_end\d+:', body, re.M).start()]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.') and not l.strip().endswith(':')]


def regs(tok):
    out = []
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', tok):
        out += list(range(int(a), int(b) + 1))
    tok = re.sub(r'v\[\d+:\d+\]', '', tok)
    out += [int(x) for x in re.findall(r'\bv(\d+)\b', tok)]
    return out


events = []  # (line, reg, 'w'/'r')
for i, l in enumerate(lines):
    parts = l.split(None, 1)
    if len(parts) < 2:
        continue
    op, args = parts
    ops = [a.strip() for a in args.split(',')]
    if not ops:
        continue
    stores = op.startswith('global_store') or op.startswith('ds_write') or op.startswith('scratch_store') or op.startswith('s_') or op.startswith('v_cmp') or op.startswith('buffer_store')
    for k, a in enumerate(ops):
        for r in regs(a):
            events.append((i, r, 'w' if (k == 0 and not stores) else 'r'))
# live intervals
last_w = {}
last_r = {}
iv = []
for i, r, k in events:
    if k == 'w':
        if r in last_w and r in last_r and last_r[r] >= last_w[r]:
            iv.append((last_w[r], last_r[r]))
        last_w[r] = i
        last_r.pop(r, None) if False else None
    else:
        last_r[r] = i
        if r not in last_w:
            last_w[r] = 0
for r in last_w:
    if r in last_r and last_r[r] >= last_w[r]:
        iv.append((last_w[r], last_r[r]))
live = [0] * (len(lines) + 1)
for a, b in iv:
    for i in range(a, b + 1):
        live[i] += 1
step = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for i in range(0, len(lines), step):
    w = lines[i:i + step]
    print(i, 'live max', max(live[i:i + step]), 'bperm', sum('ds_bpermute' in l for l in w), 'dpp', sum('dpp' in l for l in w),
          'gload', sum('global_load' in l for l in w), 'dsr', sum('ds_read' in l for l in w), 'dsw', sum('ds_write' in l for l in w),
          'rcp', sum('v_rcp' in l for l in w), 'med3', sum('v_med3' in l for l in w))
