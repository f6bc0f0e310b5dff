import re, sys
src = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^' + re.escape(name) + r':.*?s_endpgm', src, flags=re.S | re.M)
L = m.group(0).split('\n')
labels = {mm.group(1): i for i, l in enumerate(L) for mm in [re.match(r'^(\.LBB\d+_\d+):', l)] if mm}
best = None
for i, l in enumerate(L):
    mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        a, b = labels[mm.group(1)], i
        seg = L[a:b]
        mf = sum(1 for x in seg if 'v_mfma' in x)
        if mf and (best is None or (b - a) > best[1] - best[0]):
            best = (a, b)
a, b = best
seg = L[a:b]
cnt = lambda pat: sum(1 for x in seg if re.match(pat, x))
print(name[-40:], 'loop len', b - a, 'valu', cnt(r'\s+v_(?!mfma)'), 'mfma', cnt(r'\s+v_mfma'), 'salu', cnt(r'\s+s_(?!waitcnt|barrier|nop)'),
      'ds_read', cnt(r'\s+ds_read'), 'ds_write', cnt(r'\s+ds_write'), 'bufload', cnt(r'\s+buffer_load'), 'waitcnt', cnt(r'\s+s_waitcnt'), 'barrier', cnt(r'\s+s_barrier'))
if len(sys.argv) > 3:
    import collections
    c = collections.Counter(x.split()[0] for x in seg if re.match(r'\s+[vs]_', x))
    print(c.most_common(25))
