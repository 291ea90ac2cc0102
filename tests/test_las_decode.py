"""LAS point records -> positions + attribute columns (core/io/LASFile.cpp:79-94,
578-632).  Records are LAS point data record formats 0-10 (LAS 1.2 - 1.4), built here with numpy structured dtypes; the
expected values are computed in this file with numpy, independently of both the oracle and the GPU."""
import numpy as np
import pytest

import oracle_lib as O

CORE = [("X", "<i4"), ("Y", "<i4"), ("Z", "<i4"), ("intensity", "<u2"), ("bits", "u1"), ("classification", "u1"),
        ("scan_angle_rank", "i1"), ("user_data", "u1"), ("point_source_id", "<u2")]
# LAS 1.4 "Point Data Record Format 6": returns byte = return number:4 | number of returns:4; flags byte = classification
# flags:4 | scanner channel:2 | scan direction:1 | edge of flight line:1; scan angle in units of 0.006 degree
CORE14 = [("X", "<i4"), ("Y", "<i4"), ("Z", "<i4"), ("intensity", "<u2"), ("returns", "u1"), ("flags", "u1"), ("classification", "u1"),
          ("user_data", "u1"), ("scan_angle", "<i2"), ("point_source_id", "<u2"), ("gps_time", "<f8")]
WAVE = [("wave_packet", "u1", 29)]
FORMATS = {0: CORE, 1: CORE + [("gps_time", "<f8")], 2: CORE + [("rgb", "<u2", 3)],
           3: CORE + [("gps_time", "<f8"), ("rgb", "<u2", 3)],
           4: CORE + [("gps_time", "<f8")] + WAVE, 5: CORE + [("gps_time", "<f8"), ("rgb", "<u2", 3)] + WAVE,
           6: CORE14, 7: CORE14 + [("rgb", "<u2", 3)], 8: CORE14 + [("rgb", "<u2", 3), ("nir", "<u2")],
           9: CORE14 + WAVE, 10: CORE14 + [("rgb", "<u2", 3), ("nir", "<u2")] + WAVE}
SIZES = {0: 20, 1: 28, 2: 26, 3: 34, 4: 57, 5: 63, 6: 30, 7: 36, 8: 38, 9: 59, 10: 67}


def make_records(rng, n, fmt, extra):
    dt = np.dtype(FORMATS[fmt] + ([("extra", "u1", extra)] if extra else []))
    assert dt.itemsize == SIZES[fmt] + extra
    r = np.zeros(n, dtype=dt)
    for ax in "XYZ":
        r[ax] = rng.integers(-2**31, 2**31 - 1, n, endpoint=True)
    ext = np.array([0, -2**31, 2**31 - 1, 123456])[:n]      # extremes: clamped into the header box
    r["X"][:len(ext)] = ext
    r["intensity"] = rng.integers(0, 65535, n, endpoint=True)
    if fmt >= 6:
        r["returns"] = rng.integers(0, 255, n, endpoint=True)
        r["flags"] = rng.integers(0, 255, n, endpoint=True)
        r["scan_angle"] = rng.integers(-32768, 32767, n, endpoint=True)
        r["scan_angle"][:6] = np.array([0, 83, 84, -84, 30000, -30000])[:min(n, 6)]   # around a rounding step, the clamps
    else:
        r["bits"] = rng.integers(0, 255, n, endpoint=True)
        r["scan_angle_rank"] = rng.integers(-128, 127, n, endpoint=True)
    r["classification"] = rng.integers(0, 255, n, endpoint=True)
    r["user_data"] = rng.integers(0, 255, n, endpoint=True)
    r["point_source_id"] = rng.integers(0, 65535, n, endpoint=True)
    if "gps_time" in dt.names:
        r["gps_time"] = rng.random(n) * 1e9
    if "rgb" in dt.names:
        r["rgb"] = rng.integers(0, 65535, (n, 3), endpoint=True)
    if "nir" in dt.names:
        r["nir"] = rng.integers(0, 65535, n, endpoint=True)
    if "wave_packet" in dt.names:
        r["wave_packet"] = rng.integers(0, 255, (n, 29), endpoint=True)
    if extra:
        r["extra"] = rng.integers(0, 255, (n, extra), endpoint=True)
    return r


LAYOUT = dict(scale=[1e-3, 1e-3, 2e-3], offset=[500000.0, -1234.5, 0.25],
              bmin=[500000.0 - 1.5e6, -1234.5 - 1.0e6, -3.0e6], bmax=[500000.0 + 1.5e6, -1234.5 + 2.0e6, 4.0e6])


def expected(r):
    xyz = np.empty((len(r), 3))
    for k, ax in enumerate("XYZ"):
        p = LAYOUT["offset"][k] + r[ax].astype(np.float64) * LAYOUT["scale"][k]
        xyz[:, k] = np.minimum(LAYOUT["bmax"][k], np.maximum(LAYOUT["bmin"][k], p))
    names = r.dtype.names
    if "returns" in names:
        # what LASzip's raw reader leaves in the legacy fields of a laszip_point for a LAS 1.4 record
        # (LASreadItemRaw_POINT14_LE::read): more than 7 returns saturate, classes above 31 do not fit the 5-bit field,
        # the scan angle becomes whole degrees (float product, rounded half away from zero, clamped to a signed byte)
        rn, nor = (r["returns"] & 15).astype(np.int64), (r["returns"] >> 4).astype(np.int64)
        ret = np.where(nor > 7, np.where(rn > 6, np.where(rn >= nor, 7, 6), rn), rn & 7)
        nret = np.where(nor > 7, 7, nor)
        deg = np.float32(0.006) * r["scan_angle"].astype(np.float32)
        q = np.where(deg >= 0, np.trunc(deg + np.float32(0.5)), np.trunc(deg - np.float32(0.5))).astype(np.int64)
        legacy = {"return_number": ret, "number_of_returns": nret, "scan_direction_flag": (r["flags"] >> 6) & 1,
                  "edge_of_flight_line": (r["flags"] >> 7) & 1, "classification": np.where(r["classification"] < 32, r["classification"], 0),
                  "scan_angle_rank": np.clip(q, -128, 127)}
    else:
        legacy = {"return_number": r["bits"] & 7, "number_of_returns": (r["bits"] >> 3) & 7, "scan_direction_flag": (r["bits"] >> 6) & 1,
                  "edge_of_flight_line": (r["bits"] >> 7) & 1, "classification": r["classification"] & 31,
                  "scan_angle_rank": r["scan_angle_rank"]}
    a = {
        "intensity": r["intensity"], **legacy, "user_data": r["user_data"],
        "point_source_id": r["point_source_id"],
        "gps_time": r["gps_time"] if "gps_time" in names else np.zeros(len(r)),
        "rgb": (r["rgb"] >> 8).astype(np.uint8) if "rgb" in names else np.zeros((len(r), 3), np.uint8),
    }
    return xyz, a


@pytest.mark.parametrize("fmt,extra", [(0, 0), (1, 0), (2, 0), (3, 0), (3, 6), (0, 2), (4, 0), (5, 0), (6, 0), (7, 0), (8, 0), (9, 0), (10, 0),
                                       (6, 4), (10, 3)])
def test_oracle_decodes_las_records(fmt, extra):
    rng = np.random.default_rng(fmt * 10 + extra)
    r = make_records(rng, 1000, fmt, extra)
    xyz, attrs = O.las_decode(r.view(np.uint8), len(r), LAYOUT["scale"], LAYOUT["offset"], LAYOUT["bmin"], LAYOUT["bmax"], fmt,
                              r.dtype.itemsize)
    want_xyz, want = expected(r)
    assert np.array_equal(xyz, want_xyz)
    assert (xyz[:, 0] == LAYOUT["bmin"][0]).any() and (xyz[:, 0] == LAYOUT["bmax"][0]).any()   # clamping happened
    for k, v in want.items():
        assert np.array_equal(attrs[k], v.astype(attrs[k].dtype)), k


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,extra,n", [(0, 0, 1), (1, 0, 255), (2, 0, 256), (3, 0, 100_003), (3, 6, 5000), (2, 80, 777), (0, 2, 4097),
                                         (4, 0, 3000), (5, 2, 3001), (6, 0, 100_001), (7, 0, 4096), (8, 0, 257), (9, 0, 1000), (10, 0, 65_537),
                                         (6, 4, 999), (10, 40, 513)])
def test_gpu_las_decode_matches_oracle(fmt, extra, n):
    import torch
    import schwarzwald_amd as swz
    rng = np.random.default_rng(100 + fmt * 10 + extra)
    r = make_records(rng, n, fmt, extra)
    want_xyz, want = O.las_decode(r.view(np.uint8), n, LAYOUT["scale"], LAYOUT["offset"], LAYOUT["bmin"], LAYOUT["bmax"], fmt,
                                  r.dtype.itemsize)
    dev = torch.device("cuda:0")
    ctx = swz.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)  # ordered with the torch kernels that fill the buffers
    d_rec = torch.from_numpy(r.view(np.uint8).copy()).to(dev)
    d_xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_attr = {}
    for name in O.LAS_ATTRIBUTES:
        idx, dt, width = O.ATTRIBUTES[name]
        d_attr[name] = torch.full((n, width) if width > 1 else (n,), 77, dtype=getattr(torch, np.dtype(dt).name), device=dev)
    ctx.las_decode_device(d_rec.data_ptr(), n, LAYOUT["scale"], LAYOUT["offset"], LAYOUT["bmin"], LAYOUT["bmax"], fmt, r.dtype.itemsize,
                          d_xyz.data_ptr(), {k: v.data_ptr() for k, v in d_attr.items()})
    torch.cuda.synchronize()
    assert np.array_equal(d_xyz.cpu().numpy(), want_xyz)
    for k in O.LAS_ATTRIBUTES:
        assert np.array_equal(d_attr[k].cpu().numpy(), want[k]), k
    # only some columns, no positions
    part = {"intensity": torch.zeros(n, dtype=torch.uint16, device=dev)}
    ctx.las_decode_device(d_rec.data_ptr(), n, LAYOUT["scale"], LAYOUT["offset"], LAYOUT["bmin"], LAYOUT["bmax"], fmt, r.dtype.itemsize,
                          None, {k: v.data_ptr() for k, v in part.items()})
    torch.cuda.synchronize()
    assert np.array_equal(part["intensity"].cpu().numpy(), want["intensity"])
    with pytest.raises(swz.SwzError):
        ctx.las_decode_device(d_rec.data_ptr(), n, LAYOUT["scale"], LAYOUT["offset"], LAYOUT["bmin"], LAYOUT["bmax"], 11, 80, d_xyz.data_ptr())
    with pytest.raises(swz.SwzError):
        ctx.las_decode_device(d_rec.data_ptr(), n, LAYOUT["scale"], LAYOUT["offset"], LAYOUT["bmin"], LAYOUT["bmax"], 3, 30, d_xyz.data_ptr())
    with pytest.raises(swz.SwzError):
        ctx.las_decode_device(d_rec.data_ptr(), n, LAYOUT["scale"], LAYOUT["offset"], LAYOUT["bmin"], LAYOUT["bmax"], 10, 66, d_xyz.data_ptr())
    ctx.close()


@pytest.mark.gpu
def test_gpu_las_records_to_node_files(tmp_path):
    """The whole chain on the device: LAS records -> decode -> tile -> node lists -> payload gather -> BIN node
    files, against the oracle doing the same steps one after the other."""
    import os
    import torch
    import schwarzwald_amd as swz
    rng = np.random.default_rng(4242)
    n = 120_000
    r = make_records(rng, n, 3, 0)
    for ax in "XYZ":
        r[ax] = rng.integers(0, 2**30, n)            # SURVEY.md section 8(d) "LAS variant": uniform int32 in [0, 2^30)
    scale, offset = [1e-3] * 3, [0.0] * 3
    bmin, bmax = [0.0] * 3, [2**30 * 1e-3] * 3
    xyz, attrs = O.las_decode(r.view(np.uint8), n, scale, offset, bmin, bmax, 3, 34, ["rgb", "intensity"])
    spacing = O.spacing_from_diagonal(bmin, bmax, 250)
    o = O.tile(xyz, bmin, bmax, O.GRID_CENTER, 3000, spacing)
    import test_bin_persistence as TB
    d_orc, d_gpu = tmp_path / "orc", tmp_path / "gpu"
    d_orc.mkdir()
    d_gpu.mkdir()
    want = TB._oracle_node_files(str(d_orc), xyz, attrs, o)

    dev = torch.device("cuda:0")
    ctx = swz.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)  # ordered with the torch kernels that fill the buffers
    d_rec = torch.from_numpy(r.view(np.uint8).copy()).to(dev)
    d_xyz = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_attr = {"rgb": torch.empty((n, 3), dtype=torch.uint8, device=dev), "intensity": torch.empty(n, dtype=torch.uint16, device=dev)}
    ctx.las_decode_device(d_rec.data_ptr(), n, scale, offset, bmin, bmax, 3, 34, d_xyz.data_ptr(), {k: v.data_ptr() for k, v in d_attr.items()})
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    level = torch.empty(n, dtype=torch.int8, device=dev)
    order = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.tile_device(d_xyz.data_ptr(), n, bmin, bmax, swz.TileParams(sampler=swz.GRID_CENTER, max_points_per_node=3000, spacing_at_root=spacing),
                    keys.data_ptr(), perm.data_ptr(), level.data_ptr())
    nodes = ctx.build_node_lists_device(keys.data_ptr(), level.data_ptr(), n, order.data_ptr())
    out_xyz = torch.empty_like(d_xyz)
    out_attr = {k: torch.empty_like(v) for k, v in d_attr.items()}
    ctx.gather_payload_device(perm.data_ptr(), order.data_ptr(), n, d_xyz.data_ptr(), {k: v.data_ptr() for k, v in d_attr.items()},
                              out_xyz.data_ptr(), {k: v.data_ptr() for k, v in out_attr.items()})
    torch.cuda.synchronize()
    ctx.bin_persist_nodes(str(d_gpu), nodes, out_xyz.cpu().numpy(), {k: v.cpu().numpy() for k, v in out_attr.items()})
    ctx.close()
    got = sorted(os.listdir(d_gpu))
    assert got == sorted(f + ".bin" for f in want)
    for f in got:
        assert open(d_gpu / f, "rb").read() == open(d_orc / f, "rb").read(), f
