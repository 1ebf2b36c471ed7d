import sys
sys.path.insert(0,'.')
from gkr_amd import Context
from oracle import cdense
import random
P=cdense.P if hasattr(cdense,'P') else None
from oracle.field import P
c=Context(0); c.set_transcript(0)
rng=random.Random(1)
for n in (2,3,5,8,12):
    t=[rng.randrange(P) for _ in range(1<<n)]
    print(n, c.prove_sumcheck(t,n)==cdense.sumcheck_mle(t,n), flush=True)
