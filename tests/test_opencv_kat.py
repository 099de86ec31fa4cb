"""Hand-derived known answers for the OpenCV stages of the path (tests/golden/opencv_kat.json, derivations in
tests/golden/make_opencv_kat.py): the C oracle and its numpy mirror here, the HIP path in the gpu-marked test."""
import json
import os

import numpy as np
import pytest

from oracle import texture_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "opencv_kat.json")) as f:
    CASES = json.load(f)["cases"]


def _ids():
    return [f"{i}-{c['kind']}" for i, c in enumerate(CASES)]


@pytest.fixture(scope="module")
def co():
    return O.COracle()


def _gray_img(c):
    return np.array(c["gray"], dtype=np.uint8)


@pytest.mark.parametrize("case", CASES, ids=_ids())
def test_oracle_and_mirror_give_the_hand_derived_answers(co, case):
    k = case["kind"]
    if k == "gray":
        bgr = np.array(case["bgr"], dtype=np.uint8).reshape(1, -1, 3)
        want = np.array(case["expect"], dtype=np.uint8).reshape(1, -1)
        assert np.array_equal(co.gray(bgr, case["shift"]), want)
        assert np.array_equal(O.np_gray(bgr, case["shift"]), want)
    elif k == "gauss5":
        g = _gray_img(case)
        want = np.array(case["expect"], dtype=np.uint8)
        for blur in (co.gauss5(g), O.np_gauss5(g)):
            assert np.array_equal(blur, want)
            assert int(np.abs(g.astype(np.int64) - blur.astype(np.int64)).sum()) == case["absdiff_sum"]
        for hp in (co.highpass(g), O.np_highpass(g)):
            assert int(hp.astype(np.int64).sum()) == case["hp_sum"]
            assert int((hp.astype(np.int64) ** 2).sum()) == case["hp_sumsq"]
    elif k == "laplacian":
        g = _gray_img(case)
        want = (case["lap_sum"], case["lap_sumsq"], case["gray_sumsq"])
        assert tuple(int(v) for v in co.quality_sums(g)) == want
        assert tuple(int(v) for v in O.np_quality_sums(g)) == want
    elif k == "canny":
        g = _gray_img(case)
        want = np.array(case["edges"], dtype=np.uint8)
        got_c = co.canny(g, case["low"], case["high"])
        got_c = got_c[0] if isinstance(got_c, tuple) else got_c
        assert np.array_equal(got_c, want), (got_c, want)
        assert np.array_equal(O.np_canny(g, case["low"], case["high"])[0], want)
    else:
        raise AssertionError(k)


@pytest.mark.gpu
def test_hip_path_gives_the_hand_derived_answers():
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    try:
        for case in CASES:
            k = case["kind"]
            p = ctx.default_params()
            if k == "gray":
                p.gray_shift = case["shift"]
                bgr = np.array(case["bgr"], dtype=np.uint8).reshape(1, -1, 3)
                # as a one-row frame and as a column of one-pixel rows (both loaders of the gray conversion)
                for frame in (bgr, np.ascontiguousarray(bgr.reshape(-1, 1, 3))):
                    gray, _, _ = ctx.debug_planes(frame, p)
                    assert gray.reshape(-1).tolist() == case["expect"], case["derivation"]
                continue
            g = _gray_img(case)
            frame = np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))  # B = G = R = v converts to v
            if k == "gauss5":
                m = ctx.analyze_frames([frame], p)[0]
                assert (m.absdiff_sum, m.hp_sum, m.hp_sumsq) == (case["absdiff_sum"], case["hp_sum"], case["hp_sumsq"]), case["derivation"]
                assert m.gray_sum == int(g.sum())
            elif k == "laplacian":
                p.quality_stats = 1
                m = ctx.analyze_frames([frame], p)[0]
                assert (m.lap_sum, m.lap_sumsq, m.gray_sumsq) == (case["lap_sum"], case["lap_sumsq"], case["gray_sumsq"])
            elif k == "canny":
                p.canny_low, p.canny_high = case["low"], case["high"]
                want = np.array(case["edges"], dtype=np.uint8)
                _, _, edges = ctx.debug_planes(frame, p)
                assert np.array_equal(edges, want), (case["derivation"], edges)
                for mode in ("dense", "sparse"):
                    ctx.set_canny_mode(mode)
                    m = ctx.analyze_frames([frame], p)[0]
                    assert m.edge_count == int((want > 0).sum()), (mode, case["derivation"])
                ctx.set_canny_mode("auto")
    finally:
        ctx.close()
