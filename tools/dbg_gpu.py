import sys; sys.path.insert(0,'.')
import zlib
import zsc_amd as z
from oracle.oracle_py import Oracle
from zsc_amd import corpus
O=Oracle()
def t(tag, comp, cap, wb=15):
    g = z.uncompress2(comp,cap,wb); o = O.uncompress(comp,cap,wb)
    ok = g == o
    first = next((i for i,(a,b) in enumerate(zip(g[1],o[1])) if a!=b), None)
    print(tag, "OK" if ok else "DIFF", "gpu", g[0], len(g[1]), g[2], "orc", o[0], len(o[1]), o[2], "firstdiff", first, g[1][max(0,(first or 0)-4):(first or 0)+8] if first is not None else "", o[1][max(0,(first or 0)-4):(first or 0)+8] if first is not None else "")
t("fhcrc", bytes(int(x,16) for x in "1f 8b 8 2 0 0 0 0 0 0 1d 26 3 0 0 0 0 0 0 0 0 0".split()), 0, 47)
for s in (b"a", b"abc", b"hello hello", b"abcabcabcabcabcabcabc", b"a"*300, bytes(range(256))*3):
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED); c = co.compress(s)+co.flush(); t(("fixed",len(s)), c, len(s))
    t(("dyn",len(s)), zlib.compress(s,6), len(s))
    t(("stored",len(s)), zlib.compress(s,0), len(s))
for n in (100, 1000, 5000, 40000):
    d = corpus.make_buffer("text", n, 3); t(("text",n), O.compress(d,6)[1], n)
    d = corpus.make_buffer("random", n, 3); t(("random",n), O.compress(d,6)[1], n)
