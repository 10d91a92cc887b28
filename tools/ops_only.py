"""every op of bench.py's `roofline_ops` leg launched ONCE, each preceded by a marker kernel whose grid size encodes the op's key
(three_nn_weights_kernel on 256*(k+101) elements; the geometry itself launches that kernel on 4096 / 16384 / 262144): the target of the PMC passes of tools/pmc_ops.sh.  Prints the key table."""
import ctypes, json, sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import _lib as L
from gspn_amd.fea_extractor import pn2_geometry
dev = torch.device('cuda', 0)
xyz_np, _ = bench.synth(8, 32768, 0)
xyz = torch.from_numpy(xyz_np).to(dev)
geo = pn2_geometry(xyz)
keys = []
mark_in = torch.ones(256 * 200 * 3, device=dev)
mark_out = torch.empty_like(mark_in)


def timer(key, fn):
    fn()                                         # warm (un-marked: counted under the previous key's tail, dropped by the summary)
    torch.cuda.synchronize()
    k = len(keys)
    keys.append(key)
    L.check(L.lib().gspn_three_nn_weights(256 * (k + 101), L.ptr(mark_in), L.ptr(mark_out), L.stream()), "marker")
    fn()
    L.check(L.lib().gspn_three_nn_weights(256 * 100, L.ptr(mark_in), L.ptr(mark_out), L.stream()), "end marker")     # closes the bracket
    torch.cuda.synchronize()
    return 1.0


bench.ops_roofline(xyz, geo, dev, timer=timer)
assert len(keys) < 99
print("KEYS " + json.dumps(keys))
