"""Order of memory loads and waits in a kernel's ISA: `python tools/dbg/isa_trips.py file.s kernel_substring`
L = vector load, S = scalar load, D = LDS read, st = store, W(n) = s_waitcnt vmcnt(n), w = s_waitcnt lgkmcnt, B = barrier, br = branch"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):\s*; @', s, re.M):
    if sys.argv[2] not in m.group(1): continue
    e = s.find('.end_amdhsa_kernel', m.end())
    out = []
    for l in s[m.end():e].split('\n'):
        l = l.strip()
        t = None
        if re.match(r'(global_load|buffer_load|flat_load)', l): t = 'L'
        elif re.match(r'(s_load|s_buffer_load)', l): t = 'S'
        elif re.match(r'ds_read', l): t = 'D'
        elif re.match(r'(global_store|buffer_store)', l): t = 'st'
        elif re.match(r'global_atomic', l): t = 'AT'
        elif l.startswith('s_waitcnt'):
            v = re.search(r'vmcnt\((\d+)\)', l); k = 'lgkmcnt' in l
            t = ('W(%s)' % v.group(1) if v else '') + ('w' if k else '')
        elif l.startswith('s_barrier'): t = 'B'
        elif re.match(r's_cbranch|s_branch', l): t = 'br'
        elif re.match(r'v_mfma', l): t = 'M'
        elif re.match(r'\.LBB', l): t = '|'
        if t:
            if out and out[-1][0] == t: out[-1][1] += 1
            else: out.append([t, 1])
    print(m.group(1)[:80])
    print(' '.join(t if n == 1 else '%s*%d' % (t, n) for t, n in out))
