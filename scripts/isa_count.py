#!/usr/bin/env python3
"""Static instruction-class counts of one kernel in a hipcc -S listing (whole body, all paths)."""
import collections, re, sys
s = open(sys.argv[1]).read()
for pat in sys.argv[2:]:
    m = re.search(r'^(_ZN2fc\w*%s\w*):' % re.escape(pat), s, re.M)
    st = m.end(); en = s.index('s_endpgm', st)
    c = collections.Counter(); other = collections.Counter()
    for line in s[st:en].split('\n'):
        t = line.strip().split(' ')[0] if line.strip() else ''
        if not t or t[0] in '.;' or t.endswith(':'): continue
        for pre in ('v_pk', 'ds_read', 'ds_write', 'buffer_load', 'buffer_store', 'global_load', 'global_store', 'scratch', 's_waitcnt', 's_barrier'):
            if t.startswith(pre): c[pre] += 1; break
        else:
            if t.startswith('v_'): c['valu_other'] += 1; other[t] += 1
            elif t.startswith('s_'): c['salu'] += 1
            else: c[t] += 1
    print(m.group(1)[:70], dict(c)); print('   ', other.most_common(10))
