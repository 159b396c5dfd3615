"""the JSON stage files of the reference (data copies under tests/golden/jsons/, the reference's
own jsons/ directory) through the drop-in loaders (mocapv2_amd/lib/Helpers.py, reference lib/Helpers.py:13-40,
282-291) and the writers (mocapv2_amd/calibrate.py, reference CalculateCameraPoses.py:236-281): schemas, round
trips, and -- on the GPU -- the known-answer run K1 end to end from the files."""
import json
import os

import numpy as np
import pytest

from mocapv2_amd import calibrate as cal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
JSONS = os.path.join(GOLDEN, "jsons")


@pytest.fixture
def helpers(monkeypatch):
    """The drop-in lib.Helpers with its module globals reset, working directory = tests/golden (so that the
    reference's hard-coded ./jsons/... paths resolve to the bundled data)."""
    from mocapv2_amd.lib import Helpers as H
    monkeypatch.chdir(GOLDEN)
    monkeypatch.setattr(H, "camera_params", None)
    monkeypatch.setattr(H, "Fs", [])
    monkeypatch.setattr(H, "camera_params_path", "./jsons/camera-params-in.json")
    return H


def test_bundled_stage_files_have_the_documented_schemas():
    names = sorted(os.listdir(JSONS))
    assert len(names) == 14
    for stage in ("before_ba_", "after_ba_", "after_origin_", "after_floor_"):
        with open(os.path.join(JSONS, stage + "extrinsics.json")) as f:
            ext = json.load(f)
        assert len(ext) >= 2
        for pose in ext:
            assert set(pose) == {"R", "t"}
            assert np.asarray(pose["R"]).shape == (3, 3) and np.asarray(pose["t"]).shape == (3,)
    with open(os.path.join(JSONS, "camera-params-in.json")) as f:
        prm = json.load(f)
    assert len(prm) == 2
    for p in prm:
        assert np.asarray(p["intrinsic_matrix"]).shape == (3, 3) and np.asarray(p["distortion_coef"]).size == 5
    with open(os.path.join(JSONS, "fundamentals.json")) as f:
        Fs = json.load(f)
    assert np.asarray(Fs).shape == (2, 3, 3) and Fs[0] == Fs[1]  # the doubled append of CalculateCameraPoses.py:190-191
    with open(os.path.join(JSONS, "image_points.json")) as f:
        assert np.asarray(json.load(f)).shape == (54, 2, 2)
    with open(os.path.join(JSONS, "after_ba_objects.json")) as f:
        assert np.asarray(json.load(f)).shape == (54, 3)


def test_loaders_read_the_reference_files(helpers, capsys):
    H = helpers
    poses, count = H.get_extrinsics()  # default path ./jsons/after_ba_extrinsics.json
    assert count == 2 and isinstance(poses[0]["R"], np.ndarray) and poses[0]["R"].shape == (3, 3)
    assert poses[1]["t"].shape == (3,) and np.array_equal(poses[0]["R"], np.eye(3))
    for stage, n in (("before_ba_", 2), ("after_origin_", 6), ("after_floor_", 2)):
        p, c = H.get_extrinsics(f"./jsons/{stage}extrinsics.json")
        assert c == n == len(p) and all(q["R"].shape == (3, 3) and q["t"].shape == (3,) for q in p)
    prm = H.read_camera_params()
    assert isinstance(prm, np.ndarray) and prm.dtype == object and prm.shape == (2,)
    assert H.camera_params is prm and H.read_camera_params() is None  # second call: already loaded, returns nothing (:33-40)
    assert prm[0]["intrinsic_matrix"][0][0] > 1000 and len(prm[1]["distortion_coef"]) == 5
    H.read_fundamental_matrix()
    assert np.asarray(H.Fs).shape == (2, 3, 3)
    out = capsys.readouterr().out
    assert "Camera params loaded" in out and "Fundamental matrix loaded" in out
    # the module globals stay assignable (callers and the fixtures set them directly)
    H.Fs = [np.eye(3).tolist()]
    H.read_fundamental_matrix()
    assert len(H.Fs) == 1


def test_get_points_transposes_to_camera_major():
    pts = cal.get_points(os.path.join(JSONS, "image_points.json"))
    assert pts.shape == (2, 54, 2)
    with open(os.path.join(JSONS, "image_points.json")) as f:
        raw = np.array(json.load(f))
    assert np.array_equal(pts[1, 7], raw[7, 1])


@pytest.mark.parametrize("stage", ["before_ba_", "after_ba_", "after_origin_", "after_floor_"])
def test_save_extrinsics_round_trips_the_bundled_files(helpers, tmp_path, stage):
    H = helpers
    poses, count = H.get_extrinsics(f"./jsons/{stage}extrinsics.json")
    name = cal.save_extrinsics(poses, stage, directory=str(tmp_path))
    assert os.path.basename(name) == f"{stage}extrinsics.json"
    with open(name) as f, open(os.path.join(JSONS, f"{stage}extrinsics.json")) as g:
        assert json.load(f) == json.load(g)  # same schema, same numbers (repr round trip of float64)
    again, n2 = H.get_extrinsics(name)
    assert n2 == count and all(np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) for a, b in zip(again, poses))
    # (3, 1) column translations, as set_floor leaves them (CalculateCameraPoses.py:318), are flattened on the way out
    cols = [{"R": p["R"], "t": p["t"].reshape(3, 1)} for p in poses]
    with open(cal.save_extrinsics(cols, "col_", directory=str(tmp_path))) as f:
        assert all(np.asarray(q["t"]).shape == (3,) for q in json.load(f))


def test_save_objects_and_fundamentals_round_trip(helpers, tmp_path, monkeypatch):
    with open(os.path.join(JSONS, "after_ba_objects.json")) as f:
        objs = json.load(f)
    name = cal.save_objects("after_ba_", np.array(objs), directory=str(tmp_path))
    with open(name) as f:
        assert json.load(f) == objs
    with open(os.path.join(JSONS, "fundamentals.json")) as f:
        Fs = json.load(f)
    os.makedirs(tmp_path / "jsons")
    fname = cal.save_fundamentals(cal.pair_fundamentals(Fs), directory=str(tmp_path / "jsons"))
    with open(fname) as f:
        assert json.load(f) == Fs  # including the duplicate entry
    # ... and the drop-in loader reads what the writer wrote
    H = helpers
    monkeypatch.chdir(tmp_path)
    H.Fs = []
    H.read_fundamental_matrix()
    assert H.Fs == Fs


@pytest.mark.gpu
def test_k1_end_to_end_from_the_stage_files(helpers, tmp_path):
    """jsons/image_points.json + after_ba_extrinsics.json + camera-params-in.json through the loaders and the HIP
    triangulation = jsons/after_ba_objects.json (the reference's own output, <= 1e-7), written back with the
    reference's writer and compared as files; then one correspondence call with the loaded fundamentals.json."""
    H = helpers
    poses, count = H.get_extrinsics()
    pts = cal.get_points("./jsons/image_points.json")            # [2][54][2]
    groups = np.transpose(pts, (1, 0, 2))                         # CalculateCameraPoses.py:243
    obj = H.triangulate_points(groups, poses)
    with open("./jsons/after_ba_objects.json") as f:
        want = np.array(json.load(f))
    assert obj.shape == want.shape == (54, 3)
    assert np.abs(obj - want).max() < 1e-7
    name = cal.save_objects("after_ba_", obj, directory=str(tmp_path))
    with open(name) as f:
        assert np.abs(np.array(json.load(f)) - want).max() < 1e-7
    err = H.calculate_reprojection_errors(groups, obj, poses)
    assert err.shape == (54,) and np.isfinite(err).all() and float(np.mean(err)) < 1000.0
    # correspondence on the first four captured pairs with the file's fundamental matrix (K2: all 54 pairs lie
    # within the 10 px cutoff of their epipolar line)
    image_points = [[list(map(int, p)) for p in pts[0][:4]], [list(map(int, p)) for p in pts[1][:4]]]
    object_points, image_points_all = H.find_point_correspondance_and_object_points(image_points, poses, 4)
    assert image_points_all.shape[1:] == (2, 2) and len(object_points) >= 1
    for g in image_points_all:
        i = [tuple(p) for p in image_points[0]].index(tuple(g[0]))
        k = [tuple(p) for p in pts[0][:4].astype(int)].index(tuple(g[0]))
        assert i == k
    # every root's own partner is among the groups' camera-1 points
    assert all(tuple(g[1]) in [tuple(p) for p in pts[1][:4].astype(int)] for g in image_points_all)
