"""Where does a round of the resident layer kernel spend its time?  Needs a library built with -DGKR_PERSIST_DEBUG
(make -C gkr_amd/csrc CXXFLAGS+=-DGKR_PERSIST_DEBUG after touching kernels.hip): block 0 stamps wall_clock64 (100 MHz)
and clock64 at its phase boundaries of the b-rounds; this prints the intervals."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gkr_amd import Context, synth
from gkr_amd import _native as N
from gkr_amd.aggregate import ProvingStep
from gkr_amd.field import as_limbs
hip = ctypes.CDLL("libamdhip64.so")
step = ProvingStep(synth.mimc7_demo_r1cs())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(2 + i, 3 + (i % 5))) for i in range(n)]))
ctx = Context(0)
for _ in range(3):
    ctx.prove_batch_raw(step.circuits[7], inputs[7])
sym = ctypes.c_void_p()
size = ctypes.c_size_t()
# the debug array lives in the library's code object
lib = N.lib()
buf = (ctypes.c_ulonglong * 4096)()
fn = getattr(lib, "gkr_debug_read_persist", None)
if fn is None:
    raise SystemExit("library built without GKR_PERSIST_DEBUG")
fn(buf)
names = {0: "start", 1: "round start", 2: "sums done", 3: "published", 4: "challenge in", 5: "b-phase done"}
prev = None
for i in range(0, 120):
    wall, tagclk = buf[2 * i], buf[2 * i + 1]
    if wall == 0:
        break
    tag, clk = tagclk >> 56, tagclk & ((1 << 56) - 1)
    if prev:
        dw, dc = (wall - prev[0]) * 10e-3, clk - prev[1]
        print("%-14s +%7.2f us  %8d cycles  (%.0f MHz)" % (names.get(tag, tag), dw, dc, dc / dw if dw else 0))
    else:
        print(names.get(tag, tag))
    prev = (wall, clk)
