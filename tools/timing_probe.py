import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from gkr_amd import Context, synth
from gkr_amd.aggregate import ProvingStep
from gkr_amd.field import as_limbs
step = ProvingStep(synth.mimc7_demo_r1cs())
n = int(sys.argv[1])
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(n)]))
ctx = Context(0)
for _ in range(3):
    step.prove_raw(ctx, inputs)
os.environ["X"]="1"
t=time.perf_counter(); ctx.prove_batch_raw(step.circuits[7], inputs[7]); print("one sub-circuit (k list %s) ms:" % step.circuits[7].get_k_list(), (time.perf_counter()-t)*1e3)
