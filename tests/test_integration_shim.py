"""INTEGRATION.md section A, executed: the reference's OWN objects for `bgt view` (oracle/_ref/{view,bgt,vcf,hts,bgzf,fmf,
kexpr,bedidx}.o -- compiled from /root/reference in the build container, shipped with the snapshot) linked with the codec
shim tests/integration/pbf_gpu.c over libbgt_hip.so instead of pbwt.o.  The reference's unmodified reader and front end
then decode every genotype on the MI355X; its output must equal the goldens the all-CPU reference produced."""
import json
import os
import subprocess

import pytest

from conftest import require_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "bgt")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
REF_OBJS = ["view", "bgt", "vcf", "hts", "bgzf", "fmf", "kexpr", "bedidx"]


@pytest.fixture(scope="module")
def refview_gpu(tmp_path_factory):
    import bgt_amd
    bgt_amd.build_library()
    objs = [require_ref(o + ".o") for o in REF_OBJS]
    exe = str(tmp_path_factory.mktemp("shim") / "refview_gpu")
    lib = os.path.join(ROOT, "bgt_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "integration", "pbf_gpu.c")] + objs +
                          ["-o", exe, "-L", lib, "-lbgt_hip", "-Wl,-rpath," + lib, "-lz", "-lm", "-lpthread"])
    return exe


def test_shim_links_against_the_reference_objects(refview_gpu):
    """(CPU) every symbol bgt.o / view.o need from pbwt.o is provided by the shim; `view` without arguments prints the
    reference's usage text"""
    p = subprocess.run([refview_gpu, "view"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Usage: bgt view" in p.stderr


# views that read genotypes through pbf_seek / pbf_read: whole cohort, subsets, groups, regions, two databases, BCF
SHIM_VIEWS = [n for n in sorted(MANIFEST["views"]) if MANIFEST["views"][n]["rc"] == 0 and
              not any(a in ("-d", "-M") for a in MANIFEST["views"][n]["args"])]


@pytest.mark.gpu
@pytest.mark.parametrize("name", SHIM_VIEWS)
def test_reference_front_end_on_the_hip_codec(refview_gpu, name):
    v = MANIFEST["views"][name]
    p = subprocess.run([refview_gpu, "view"] + v["args"] + v["prefixes"], cwd=GOLD, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert p.stdout == open(os.path.join(GOLD, "expected", name + ".out"), "rb").read()
