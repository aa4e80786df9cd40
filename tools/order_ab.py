import os, sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from brotli_rs_amd import brx
import brx_knobs  # noqa: E402
G='/root/repo/tests/golden'
man=json.load(open(G+'/config5/manifest.json'))['streams']
big=[open(G+'/config5/%s.compressed'%e['name'],'rb').read() for e in man]
small=open(G+'/data/alice29.txt.compressed','rb').read()
streams=[small]*4096+[big[i%4] for i in range(64)]
caps=[152089]*4096+[1<<20]*64
ctx=brx_knobs.context(0)
for k in range(3):
    t=time.time(); outs,st,ol=ctx.decode_batch(streams,caps,timing=True) if 'timing' in brx.Context.decode_batch.__code__.co_varnames else ctx.decode_batch(streams,caps); dt=time.time()-t
    print('NO_ORDER' if os.environ.get('BRX_NO_ORDER') else 'ordered', 'wall %.1f ms'%(dt*1e3), 'kernel', ctx.last_timing_ms(1) if hasattr(ctx,'last_timing_ms') else None, int(st.any()))
