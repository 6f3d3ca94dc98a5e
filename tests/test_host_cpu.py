"""CPU: host logic that needs no GPU -- file formats (round trip + the reference reader's quirks) and
the evaluation metrics of the Python mirror against the oracle's C versions."""
import numpy as np


def test_valset_pose_descriptor_round_trip(tmp_path):
    from gloc3d_amd import gloc_io, synth
    db = [f"/data/db/{i:06d}.bin" for i in range(4)]
    q = [f"/data/q/{i:06d}.bin" for i in range(3)]
    pos = [[1, 2], [], [3]]
    p = tmp_path / "valset.txt"
    gloc_io.write_valset(p, db, q, pos)
    assert gloc_io.read_valset(p) == (db, q, pos)
    # quirks: the token before ':' is ignored; the section may be shorter than numQ
    txt = p.read_text().split("\n")
    txt[-4] = "999:1 2 "            # wrong query index + trailing space: still positional
    p.write_text("\n".join(txt[:-2]) + "\n")  # drop the last positives line
    db2, q2, pos2 = gloc_io.read_valset(p)
    assert pos2 == [[1, 2], []]
    poses = [synth.se3(10.0 * i, (i, -i, 0.5 * i), pitch_deg=2.0 * i) for i in range(5)]
    gloc_io.write_poses(tmp_path / "poses.txt", poses)
    back = gloc_io.read_poses(tmp_path / "poses.txt")
    assert all(np.abs(a - b).max() < 1e-5 for a, b in zip(poses, back))
    d = synth.descriptors_iid(3, 0, 7, 16)
    gloc_io.write_descriptors(tmp_path / "d.desc", d)
    assert (gloc_io.read_descriptors(tmp_path / "d.desc") == d).all()


def test_recall_and_pose_error_match_oracle(oracle_mod):
    from gloc3d_amd import loop_detector as ld, synth
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 50, (30, 20)).astype(np.uint64)
    pos = [list(rng.integers(0, 50, rng.integers(0, 4))) for _ in range(30)]
    rec, failed = ld.recognition_recalls(idx, pos)
    valid, orec = oracle_mod.recall_at(idx, pos)
    assert np.allclose(rec, orec) and valid == sum(1 for p in pos if len(p))
    a, b = synth.se3(10.0, (1, 2, 3)), synth.se3(12.5, (1.5, 2, 3))
    er, ep = ld.pose_error(a, b)
    oer, oep = oracle_mod.pose_error(a, b)
    assert abs(er - oer) < 1e-3 and abs(ep - oep) < 1e-6


def _reference_nclt_loop(raw):
    """Literal emulation of read_lidar_data_nclt (registration/global_registration.cpp:181-209): eof() is tested
    BEFORE the five reads; a read that hits the end of the file stores what it got, sets eof + fail, and later
    reads of the iteration do nothing; the point is pushed whatever happened."""
    pos, eof, failed = 0, False, False
    x = y = z = i = l = 0          # (uninitialised upstream; only matters for an empty file)
    out = []
    scaling, offset = np.float32(0.005), np.float32(-100.0)
    while True:
        if eof:
            break
        fields = []
        for width in (2, 2, 2, 1, 1):
            if failed:
                fields.append(None)
                continue
            got = raw[pos:pos + width]
            pos += len(got)
            if len(got) < width:
                eof = failed = True
            fields.append(bytes(got))
        old = [x.to_bytes(2, "little"), y.to_bytes(2, "little"), z.to_bytes(2, "little"), bytes([i]), bytes([l])]
        new = [(f + o[len(f):]) if f is not None else o for f, o in zip(fields, old)]
        x, y, z = (int.from_bytes(b, "little") for b in new[:3])
        i, l = new[3][0], new[4][0]
        out.append((np.float32(x) * scaling + offset, np.float32(y) * scaling + offset,
                    np.float32(z) * scaling + offset, np.float32(i)))
    return np.array(out, np.float32).reshape(-1, 4)


def test_nclt_scans_even_record_count_and_reader_quirk(tmp_path):
    """an NCLT file with an even number of 8-byte records is a multiple of 16 bytes and was read as
    KITTI floats.  The format is decided by content (or named by the caller); the reader reproduces the
    reference's loop, including its duplicated last point; the C++ host mirror returns the same bits."""
    import subprocess
    from gloc3d_amd import gloc_io, synth
    w = synth.make_world(7)
    scan = synth.lidar_scan(w, synth.se3(3.0, (1.0, -2.0, 0.0)), seed=3, n_az=90)      # ~5k points, x y z i
    n = scan.shape[0] - scan.shape[0] % 2                                             # even: bytes % 16 == 0
    scan = scan[:n]
    f_nclt, f_kitti, f_odd = tmp_path / "a.nclt.bin", tmp_path / "a.kitti.bin", tmp_path / "odd.bin"
    gloc_io.write_lidar_nclt(f_nclt, scan[:, :3], intensity=(scan[:, 3] * 255).astype(np.uint8))
    synth.write_kitti_bin(f_kitti, scan)
    assert f_nclt.stat().st_size == 8 * n and f_nclt.stat().st_size % 16 == 0
    assert not gloc_io.looks_like_kitti(f_nclt) and gloc_io.looks_like_kitti(f_kitti)
    got = gloc_io.read_lidar_any(f_nclt)                         # auto: by content
    assert got.shape == (n + 1, 4) and (got[-1] == got[-2]).all()                     # the reference's extra point
    assert np.abs(got[:n, :3] - scan[:, :3]).max() <= 0.0025 + 1e-5                   # 5 mm quantisation
    assert (got[:n, 3] == (scan[:, 3] * 255).astype(np.uint8)).all()
    assert (gloc_io.read_lidar_any(f_kitti) == scan).all()
    raw = f_nclt.read_bytes()
    f_odd.write_bytes(raw[:8 * 101 + 3])                          # a truncated trailing record
    head = tmp_path / "head.bin"
    head.write_bytes(raw[:8 * 300])                               # whole records only
    for f, n_pts in ((head, 301), (f_odd, 102)):                  # 101 records + the truncated one (its read hits eof: no repeat)
        want = _reference_nclt_loop(f.read_bytes())
        have = gloc_io.read_lidar_nclt(f)
        assert have.shape == want.shape == (n_pts, 4) and (have.view(np.uint32) == want.view(np.uint32)).all()
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    assert gloc_io.read_lidar_nclt(empty).shape == (0, 4)
    # the C++ host mirror (csrc/host/gloc_io.hpp), compiled here: same sniff, same points, same bits
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "host_io_probe"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"),
                           "-I" + os.path.join(root, "gloc3d_amd", "csrc", "host"),
                           os.path.join(root, "tests", "host_io_probe.cpp"), "-o", str(exe)])
    for f, fmt, ref in ((f_nclt, "auto", got), (f_odd, "nclt", gloc_io.read_lidar_nclt(f_odd)),
                        (f_kitti, "auto", scan), (f_kitti, "kitti", scan)):
        out = subprocess.run([str(exe), str(f), fmt], capture_output=True, text=True, check=True).stdout.split("\n")
        kind, cnt = out[0].split()
        assert kind == ("kitti" if f is f_kitti else "nclt") and int(cnt) == ref.shape[0]
        pts = np.array([[np.float32(t) for t in ln.split()] for ln in out[1:1 + int(cnt)]], np.float32)
        assert (pts.view(np.uint32) == np.ascontiguousarray(ref, np.float32).view(np.uint32)).all()
