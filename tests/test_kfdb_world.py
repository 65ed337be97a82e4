"""The drop-in KeyFrameDatabase (include/KeyFrameDatabase.h + csrc/ref_adapter/KeyFrameDatabase.cc) against the REFERENCE'S OWN
src/KeyFrameDatabase.cc: tests/support/kfdb_world.cpp builds the same keyframe graph (three maps, covisibility, BowVectors from the
vocabulary's transform) in both builds and runs add / erase / clear / clearMap and the five Detect* routines; the candidate vectors
and the query / words / score fields of every keyframe after every call must be identical (scores as float bit patterns).

  golden   tests/golden/kfdb_world_ref.txt.gz = output of the reference build (oracle/_ref/ref_kfdb_world: the reference's file and
           the reference's DBoW2 vocabulary, compiled where they lie)
  CPU      the drop-in linked against the oracle-backed stub of the C-ABI: its host logic (list rules, thresholds, accumulation)
  GPU      the drop-in linked against liborbx.so: the shipped path (orbx_kfdb_sharing / orbx_kfdb_score on the resident CSR)
"""
import gzip
import os

import pytest

from tests import world_util as wu

GOLD = os.path.join(wu.ROOT, "tests", "golden")


def _golden(tmp_path):
    world = str(tmp_path / "world.bin")
    wu.write_world(world)     # also writes world.bin.voc.txt; test_matcher_world checks that the generator is deterministic
    assert open(world, "rb").read() == gzip.open(os.path.join(GOLD, "matcher_world.bin.gz")).read()
    return world, gzip.open(os.path.join(GOLD, "kfdb_world_ref.txt.gz")).read().decode()


def _records(txt):
    return [l for l in txt.splitlines() if not l.startswith("  ")]


def test_reference_build_reproduces_golden(tmp_path):
    if not os.path.exists(wu.REF_KFDB_EXE):
        pytest.skip("oracle/_ref/ref_kfdb_world not built (needs /root/reference)")
    world, gold = _golden(tmp_path)
    out = wu.run_kfdb_world(wu.REF_KFDB_EXE, world, str(tmp_path / "ref.txt"))
    assert out == gold, wu.first_difference(gold, out)
    recs = _records(gold)
    assert len(recs) == 24 and all("EXCEPTION" not in r for r in recs)
    assert sum(int(r.split("ret=")[1]) > 0 for r in recs) >= 14      # most calls return candidates; the others are the empty paths


def test_dropin_host_logic_equals_reference(tmp_path):
    world, gold = _golden(tmp_path)
    out = wu.run_kfdb_world(wu.build_kfdb_world("oracle"), world, str(tmp_path / "cpu.txt"))
    assert out == gold, wu.first_difference(gold, out)


@pytest.mark.gpu
def test_dropin_on_gpu_equals_reference(tmp_path):
    world, gold = _golden(tmp_path)
    out = wu.run_kfdb_world(wu.build_kfdb_world("orbx"), world, str(tmp_path / "gpu.txt"))
    assert out == gold, wu.first_difference(gold, out)


OTHER_WORLDS = [
    dict(rows=480, cols=640, nfeatures=700, steps=(0, 1, 3, 5), seed=7),
    dict(rows=376, cols=1241, nfeatures=1500, steps=(0, 3, 5, 8), seed=11),
]


def _other(tmp_path, kw):
    if not os.path.exists(wu.REF_KFDB_EXE):
        pytest.skip("oracle/_ref/ref_kfdb_world not built (needs /root/reference)")
    world = str(tmp_path / "world.bin")
    wu.write_world(world, **kw)
    ref = wu.run_kfdb_world(wu.REF_KFDB_EXE, world, str(tmp_path / "ref.txt"))
    assert len(_records(ref)) == 24 and "EXCEPTION" not in ref
    return world, ref


@pytest.mark.parametrize("kw", OTHER_WORLDS, ids=lambda kw: f"{kw['cols']}x{kw['rows']}-{kw['nfeatures']}")
def test_dropin_host_logic_equals_reference_on_other_worlds(tmp_path, kw):
    world, ref = _other(tmp_path, kw)
    out = wu.run_kfdb_world(wu.build_kfdb_world("oracle"), world, str(tmp_path / "cpu.txt"))
    assert out == ref, wu.first_difference(ref, out)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", OTHER_WORLDS, ids=lambda kw: f"{kw['cols']}x{kw['rows']}-{kw['nfeatures']}")
def test_dropin_on_gpu_equals_reference_on_other_worlds(tmp_path, kw):
    world, ref = _other(tmp_path, kw)
    out = wu.run_kfdb_world(wu.build_kfdb_world("orbx"), world, str(tmp_path / "gpu.txt"))
    assert out == ref, wu.first_difference(ref, out)
