#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Runs pyr_render_simple_multi with several logical ranks on ONE GPU through the in-process stand-in for
librccl (fake_rccl.cpp, loaded by libpyrite_gpu.so through PYRITE_RCCL_LIB), so that the multi-rank flow of
pyrite_amd/csrc/multi.cpp -- one host thread per rank, the status agreement, the grouped send / receive with trailers, the
abort on an error inside the group -- executes on the one-GPU box. Own process because the library binds its RCCL once.

    python tests/fake_rccl/run_cases.py            prints one JSON line {"case": "ok" | "<what went wrong>", ...}
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ["PYRITE_RCCL_LIB"] = os.path.join(HERE, "libfake_rccl.so")
os.environ.setdefault("FAKE_RCCL_TIMEOUT_MS", "5000")

import numpy as np  # noqa: E402

from pyrite_amd import scenes  # noqa: E402
from pyrite_amd._lib import PyriteGpuError  # noqa: E402

results = {}


def case(name):
    def wrap(fn):
        t = time.time()
        try:
            fn()
            results[name] = "ok"
        except Exception as e:  # noqa: BLE001 -- reported to the caller, which asserts on it
            results[name] = "%s: %s" % (type(e).__name__, e)
        results[name + "_seconds"] = round(time.time() - t, 2)
        return fn
    return wrap


world, cam, r, whole = scenes.build(scenes.c2_cornell(72, 56, 6), seed=5)
r.tile_size = 16  # 5 x 4 tiles, both edges cut
r.render(whole, cam, world)


def expect_error(ranks, fragment):
    film = r.new_film(72, 56)
    try:
        r.render_multi(film, cam, world, devices=[0] * ranks)
    except PyriteGpuError as e:
        assert fragment in str(e), "wrong error: %s" % e
        return
    raise AssertionError("the call returned PYR_OK")


def expect_film(ranks):
    film = r.new_film(72, 56)
    r.render_multi(film, cam, world, devices=[0] * ranks)
    assert np.array_equal(film.grains[..., 1], whole.grains[..., 1]), "weights differ from the single-device film"
    assert np.allclose(film.grains, whole.grains, rtol=1e-5), "film differs from the single-device film"


@case("two_ranks_equal_one")
def _():
    expect_film(2)


@case("three_ranks_equal_one")
def _():
    expect_film(3)


@case("more_ranks_than_tiles")
def _():
    keep = r.tile_size
    r.tile_size = 64  # 2 x 1 tiles for 4 ranks: two ranks have nothing to render and stay out of the gather
    try:
        one = r.new_film(72, 56)
        r.render(one, cam, world)
        film = r.new_film(72, 56)
        r.render_multi(film, cam, world, devices=[0] * 4)
        assert np.array_equal(film.grains[..., 1], one.grains[..., 1]) and np.allclose(film.grains, one.grains, rtol=1e-5)
    finally:
        r.tile_size = keep


@case("a_rank_that_fails_before_the_gather_fails_every_rank")
def _():
    os.environ["PYRITE_TEST_FAIL_RANK"] = "1"
    try:
        expect_error(3, "test switch: this rank's buffers could not be grown")  # the failing rank's own message, not "another rank failed"
    finally:
        del os.environ["PYRITE_TEST_FAIL_RANK"]
    expect_film(3)  # the communicators survived: nothing had been posted


@case("rank_zero_failing_before_the_gather")
def _():
    os.environ["PYRITE_TEST_FAIL_RANK"] = "0"
    try:
        expect_error(2, "test switch")
    finally:
        del os.environ["PYRITE_TEST_FAIL_RANK"]
    expect_film(2)


@case("a_launch_that_fails_after_the_agreement_travels_in_the_trailer")
def _():
    os.environ["PYRITE_TEST_FAIL_RENDER_RANK"] = "2"
    try:
        expect_error(3, "rank 2 could not launch its render")
    finally:
        del os.environ["PYRITE_TEST_FAIL_RENDER_RANK"]
    expect_film(3)


@case("a_film_the_kernels_flag_invalid_is_an_error")
def _():
    os.environ["PYRITE_SCHEDULER"] = "sm"
    os.environ["PYRITE_TEST_TAPE_OPS"] = "3"
    try:
        expect_error(2, "spectral tape")
    finally:
        del os.environ["PYRITE_TEST_TAPE_OPS"]
    try:
        expect_film(2)  # the words were cleared with the error
    finally:
        del os.environ["PYRITE_SCHEDULER"]


@case("a_lost_message_is_an_error_and_new_communicators_are_made")
def _():
    os.environ["FAKE_RCCL_DROP_SEND_FROM"] = "1"
    t = time.time()
    try:
        expect_error(2, "ncclGroupEnd")  # rank 0's receive never completes: the stand-in times out where RCCL would block
    finally:
        del os.environ["FAKE_RCCL_DROP_SEND_FROM"]
    assert time.time() - t < 30
    expect_film(2)  # the aborted communicators were dropped from the cache


world.close()

# ---- the plan the driver's 8-GPU run executes, at its real size: BASELINE's C3 image, 1920 x 1080 = 2040 ringed
# tile blocks, eight ranks of 255 blocks = 151 MB each -- rank 0 posts seven receives (1.06 GB in ONE group, every message
# under the 256 MiB the gather cuts at), the others one send each -- on the full 819,212-triangle mesh at 2 spp
if os.environ.get("FAKE_RCCL_SKIP_FULL_SIZE") != "1":
    W, H = 1920, 1080
    big_world, big_cam, big_r, big_whole = scenes.build(scenes.c3_mesh_in_box(W, H, 2), seed=3)
    big_r.render(big_whole, big_cam, big_world)

    @case("eight_ranks_at_full_size_c3_equal_one_launch")
    def _():
        film = big_r.new_film(W, H)
        big_r.render_multi(film, big_cam, big_world, devices=[0] * 8)
        assert np.array_equal(film.grains[..., 1], big_whole.grains[..., 1]), "weights differ from the single launch"
        assert np.allclose(film.grains[..., 0], big_whole.grains[..., 0], rtol=1e-4, atol=1e-6), "film differs from the single launch"

    @case("a_message_lost_mid_group_at_full_size_fails_every_rank_at_once")
    def _():
        os.environ["FAKE_RCCL_DROP_SEND_FROM"] = "3"  # rank 0 holds six arrived messages of seven when its group fails
        t = time.time()
        try:
            film = big_r.new_film(W, H)
            try:
                big_r.render_multi(film, big_cam, big_world, devices=[0] * 8)
            except PyriteGpuError as e:
                assert "ncclGroupEnd" in str(e) or "ncclRecv" in str(e), "wrong error: %s" % e
            else:
                raise AssertionError("the call returned PYR_OK")
        finally:
            del os.environ["FAKE_RCCL_DROP_SEND_FROM"]
        # the stand-in's own timeout is 5 s (where RCCL would block for ever); the failing rank aborts its siblings' communicators
        # as soon as it knows, so nobody sits out a second one
        assert time.time() - t < 40, "took %.0f s" % (time.time() - t)
        film = big_r.new_film(W, H)
        big_r.render_multi(film, big_cam, big_world, devices=[0] * 8)  # new communicators were made
        assert np.array_equal(film.grains[..., 1], big_whole.grains[..., 1])

    big_world.close()
print(json.dumps(results))
