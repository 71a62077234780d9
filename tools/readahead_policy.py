# usage: python tools/readahead_policy.py gpurun_out/stats.log   (a stats build with -DSQZ_STATS_SERIES prints the "S want m done stop" lines)
# offline what-if for the decoder's read-ahead depth.  Each recorded step has (want, m, done, stop).
# "potential" = tokens that could have been applied from the step's start: done if the step was
# cut by a bad lane / refused token (done < m or stop), otherwise censored at m (>= m).
import sys, numpy as np
rows=[tuple(map(int,l.split()[1:])) for l in open(sys.argv[1]) if l.startswith('S ')]
a=np.array(rows); want,m,done,stop=a.T
cut = (done < m) | (stop==1)          # the step's end is a real event position
# rebuild the event structure as a token stream: segments of clean tokens ending in an event
segs=[]; run=0
for d,c in zip(done,cut):
    run+=d
    if c: segs.append(run); run=0
if run: segs.append(run)
segs=np.array(segs); print("steps",len(a),"events",len(segs),"tokens",segs.sum(), "mean seg %.1f"%segs.mean())
R_TOK=6.9          # tokens found per read-ahead round
C_ROUND=2900+760   # cycles per round incl. hops
C_STEP=10700+1500  # round-1 figures: batched update + post/store per step
def simulate(policy, name):
    cyc=0; steps=0; rounds=0; state=policy.init()
    for s in segs:                      # s clean tokens, then an event (exact path, cost not policy dependent)
        left=s
        while True:
            w=policy.want(state)
            nr=int(np.ceil(min(w,64)/R_TOK)); got=min(64,int(nr*R_TOK))
            take=min(left,got)
            cyc+=nr*C_ROUND+C_STEP; steps+=1; rounds+=nr
            hit_event = left<=got          # the event position lies inside what was read
            state=policy.update(state,take,hit_event)
            left-=take
            if hit_event: break
    print(f"{name:34s} steps {steps:6d} rounds {rounds:6d} cycles {cyc/1e6:7.1f} M")
class Fixed:
    def __init__(s,w): s.w=w
    def init(s): return None
    def want(s,st): return s.w
    def update(s,st,take,ev): return None
class Ema:      # current: want = 2*mean+6 with avg4 EMA
    def __init__(s,mult=0.5,add=6): s.mult=mult; s.add=add
    def init(s): return 16
    def want(s,a4): return min(64,int(a4*s.mult)+s.add)
    def update(s,a4,take,ev): return a4+take-(a4>>2)
class Last:     # want from the last step's length by a table
    def __init__(s,tab): s.tab=tab
    def init(s): return 0
    def want(s,last):
        for lim,w in s.tab:
            if last<lim: return w
        return 64
    def update(s,last,take,ev): return take
simulate(Fixed(64),"always 64")
simulate(Ema(),"current (2*mean+6)")
simulate(Ema(0.375,6),"1.5*mean+6")
simulate(Ema(0.75,8),"3*mean+8")
simulate(Last([(1,8),(8,16),(24,32),(64,48)]),"last: 0->8 <8->16 <24->32 <64->48")
simulate(Last([(1,8),(8,24),(24,48)]),"last: 0->8 <8->24 <24->48 else 64")
simulate(Last([(1,16),(8,32)]),"last: 0->16 <8->32 else 64")
simulate(Last([(4,16)]),"last: <4->16 else 64")
