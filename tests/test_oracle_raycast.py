"""Oracle ray caster sanity (no GPU): a wall seen head-on renders a filled, shaded image."""
import numpy as np


def test_oracle_raycast_renders_wall(make_oracle):
    from ratsdf import synthetic
    vs = 0.02
    e = make_oracle(vs, 6 * vs, threads=4)
    f = synthetic.frame("wall", 0, scale=0.25)
    for _ in range(6):  # weights must reach 10 (voxel_tsdf.cu:312)
        e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    h, w = f["depth"].shape
    rgba, normal = e.raycast(f["intrinsics"], h, w, f["pose"], 8.0)
    hit = rgba[..., 3] == 255
    # blocks are only allocated where all 8 corners are in view, so the image border stays empty
    assert hit.mean() > 0.6 and hit[30:-20, 30:-30].all()
    assert (rgba[~hit] == 0).all() and (normal[~hit] == 0).all()
    # a fronto-parallel wall faces the camera: diffuse shading close to white where ht is low
    assert normal[hit][:, 1].mean() > 100
    # a row range of the same rendering (ratsdf_raycast_rows: the strips of multi.raycast_across_shards)
    for r0, r1 in ((0, h), (7, 23), (h - 5, h), (11, 11)):
        ra, rn = e.raycast_rows(f["intrinsics"], h, w, f["pose"], 8.0, r0, r1)
        assert np.array_equal(ra, rgba[r0:r1]) and np.array_equal(rn, normal[r0:r1])
    # nothing integrated -> nothing rendered
    e2 = make_oracle(vs, 6 * vs)
    r2, n2 = e2.raycast(f["intrinsics"], h, w, f["pose"], 8.0)
    assert not r2.any() and not n2.any()
