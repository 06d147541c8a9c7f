"""List the control flow, waits and memory instructions of one kernel in a hipcc -S listing:
tools/isa_waits.py file.s mangled-name-fragment [first_line last_line]"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(\S*%s\S*):' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():s.index('s_endpgm', m.end())].split('\n')
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
prev, cnt = None, 0
for i, l in enumerate(body):
    if not lo <= i < hi:
        continue
    t = l.strip()
    if not (t.startswith(('s_memtime', 's_barrier', 'global_', 'buffer_', 'scratch_', 's_cbranch', 'ds_', 'v_mfma')) or 's_waitcnt' in t
            or re.match(r'^\.LBB', t)):
        continue
    key = t.split()[0]
    if key == prev and key.startswith(('global_', 'ds_', 'v_mfma', 'scratch_')):
        cnt += 1
        continue
    if cnt:
        print(f"        ... x{cnt + 1}")
        cnt = 0
    print(f"{i:5d} {t[:100]}")
    prev = key
