"""Lists the memory instructions and waits of one kernel in a hipcc -S listing: python tools/isa_mem.py file.s mangled-name-substring [max]"""
import re, sys
t = open(sys.argv[1]).read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):\s*(;.*)?$", t, re.M)
i = m.start(); j = t.index('.Lfunc_end', i)
body = t[i:j].splitlines()
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 120
print(m.group(1), len(body), 'lines')
out = [(n, l.strip()) for n, l in enumerate(body) if re.search(r'global_load|global_store|s_waitcnt vmcnt|buffer_|scratch_|Loop Header|s_barrier|ds_', l)]
# collapse runs
last = None; cnt = 0; shown = 0
for n, l in out:
    key = re.sub(r'v\[?\d+(:\d+)?\]?|s\[\d+:\d+\]|offset:-?\d+|\.LBB\S+', '', l)
    if key == last: cnt += 1; continue
    if cnt: print('      ... x%d more' % cnt)
    print(n, l[:100]); last = key; cnt = 0; shown += 1
    if shown >= lim: break
