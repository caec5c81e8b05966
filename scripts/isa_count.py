"""Instruction mix of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): python isa_count.py f.s NAME"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
lab = [m for m in re.finditer(r'^(\S*%s\S*):.*$' % re.escape(name), s, re.M)]
assert lab, "kernel not found"
start = lab[0].end()
end = s.index('s_endpgm', start)
# the body may hold several s_endpgm (early exits): go to .Lfunc_end
m = re.search(r'^\.Lfunc_end\d+:', s[start:], re.M)
body = s[start:start + m.start()]
c = collections.Counter()
for l in body.split('\n'):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
tot = sum(c.values())
valu = sum(n for o, n in c.items() if o.startswith('v_'))
print("total", tot, "valu", valu, "packed", sum(n for o, n in c.items() if o.startswith('v_pk')),
      "lds", sum(n for o, n in c.items() if o.startswith('ds_')), "global", sum(n for o, n in c.items() if o.startswith('global_')))
for o, n in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"{o:28s}{n}")
