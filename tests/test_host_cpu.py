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
