"""Far back-references (older than the LDS ring) of a stream, from the oracle's BRO_TRACE: how many 128-byte lines of the stream's own
output they touch, and how many of those a per-wave cache of the last N lines would save -- per candidate ring size.  (VERDICT r3 next #3:
do alice29's far references cluster?  They do not.)"""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = "import sys; sys.path.insert(0, %r); import oracle_py; oracle_py.decode(open(sys.argv[1],'rb').read())" % os.path.join(ROOT, "tests")
for f in sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", CODE, f], env=dict(os.environ, BRO_TRACE="1"), capture_output=True, text=True).stderr
    cmds = [tuple(map(int, l.split()[1:])) for l in out.splitlines() if l.startswith("CMD")]  # pos, insert, copy, distance
    print(f, "commands", len(cmds))
    for ring in (2048, 4096, 8192, 16384):
        far=[(c[0]+c[1]-c[3], c[2]) for c in cmds if c[3] > ring]  # source start, len ; pos is command start? assume pos+insert = copy dest
        lines=0; 
        for N in (1,2,4,8):
            cache=collections.deque(maxlen=N); miss=0
            for src,ln in far:
                for line in range(src>>7, ((src+ln-1)>>7)+1):
                    if line in cache: continue
                    miss+=1; cache.append(line)
            if N==1: base=sum(((s+l-1)>>7)-(s>>7)+1 for s,l in far)
            print("  ring %5d: far copies %5d, lines touched %5d, misses with %d-line cache: %5d" % (ring, len(far), base, N, miss))
