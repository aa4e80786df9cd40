import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from brotli_rs_amd import brx
import brx_knobs  # noqa: E402
G='/root/repo/tests/golden/data/'
small=open(G+'monkey.compressed','rb').read(); exp=open(G+'monkey','rb').read()
ctx=brx_knobs.context(0)
for n in (65536, 262144):
    t=time.time(); outs,st,ol=ctx.decode_batch([small]*n,[len(exp)]*n, timing=True); dt=time.time()-t
    ok = (not st.any()) and all(o==exp for o in outs[::997])
    print(n,'streams ok',ok,'wall %.0f ms'%(dt*1e3),'kernel ms',ctx.last_timing_ms(1))
