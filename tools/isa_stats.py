"""Register counts and instruction mix of kernels in a hipcc -save-temps .s file:  python tools/isa_stats.py file.s pattern..."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pats = sys.argv[2:] or [""]
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    name, body = m.group(1), m.group(2)
    def g(k):
        r = re.search(re.escape(k) + r':\s+(\d+)', body)
        return r.group(1) if r else '-'
    meta[name] = dict(vgpr=g('.vgpr_count'), agpr=g('.agpr_count'), sgpr=g('.sgpr_count'), spill=g('.vgpr_spill_count'),
                      scratch=g('.private_segment_fixed_size'), lds=g('.group_segment_fixed_size'))
for name, md in meta.items():
    if not any(p in name for p in pats):
        continue
    i = s.find("\n" + name + ":")
    j = s.find('s_endpgm', i)
    body = s[i:j]
    c = Counter(re.findall(r'^\s+(v_\w+|s_load\w+|global_\w+|scratch_\w+|ds_\w+|buffer_\w+)', body, re.M))
    tot = sum(c.values())
    print(name[:90]); print("  ", md, "insts", tot)
    print("  ", dict(c.most_common(16)))
