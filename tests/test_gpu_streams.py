"""GeometryStream's hardware-queue test (gspn_amd/geometry.py): a stream never runs beside itself; a GeometryStream runs beside the
stream it was created for, and two of them beside each other -- whatever streams were created before (r03: with an RCCL process
group initialised, round 2's untested side stream landed on the layers' queue)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_geometry_streams_are_tested_to_run_beside_the_layers():
    from gspn_amd.geometry import GeometryStream, runs_beside
    cur = torch.cuda.current_stream()
    assert not runs_beside(cur, cur)
    junk = [torch.cuda.Stream() for _ in range(5)]          # shift the round-robin the way another library's streams would
    for s in junk:
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
    a = GeometryStream(torch.device("cuda", 0))
    b = GeometryStream(torch.device("cuda", 0), beside=[cur, a.stream])
    assert not a.shares_queue and not b.shares_queue
    assert runs_beside(cur, a.stream) and runs_beside(cur, b.stream) and runs_beside(a.stream, b.stream)
    # and the stream still does its job
    x = torch.rand(2, 4096, 3, device="cuda")
    from gspn_amd.tf_sampling import farthest_point_sample
    pend = a.submit(lambda t: farthest_point_sample(64, t), x, after=None)
    idx = pend.get(host_wait=True)
    assert torch.equal(idx, farthest_point_sample(64, x))
