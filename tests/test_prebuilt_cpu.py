"""The prebuilt config-specialised units (ns_gym_amd/prebuilt.py): `__graft_entry__.build()` compiles the BASELINE
configurations' units for the MI355X's target without a GPU, inspects them and ships them next to the library; `nsg_specialize`
finds them by key before it asks the runtime compiler.  Here: the build-time side."""
import ctypes as C
import json
import os

import pytest

from ns_gym_amd import _lib, prebuilt


@pytest.fixture(scope="module")
def units(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("prebuilt"))
    return d, prebuilt.build_all(d)


def test_every_listed_unit_is_built_and_clean(units):
    d, manifest = units
    files = sorted(f for f in os.listdir(d) if f.endswith(".hsaco"))
    assert files == sorted(manifest) and len(files) == len(prebuilt.SINGLES) + len(prebuilt.EXACT) + len(prebuilt.GROUPS) + sum(len(p[4]) for p in prebuilt.POLICIES)
    kernels = 0
    for f, m in manifest.items():
        assert m["arch"] == "gfx950:sramecc+:xnack-"
        names = set(m["kernels"])
        assert names in ({"nsg_spec_step", "nsg_spec_rollout"}, {"nsg_spec_group", "nsg_spec_group_rollout"}, {"nsg_spec_rollout_policy"}), names
        for r in m["kernels"].values():      # the rule of nsg_specialize, asserted on what ships
            assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0
            kernels += 1
    text = open(os.path.join(d, "resource_usage.txt")).read()
    assert text.count("vgpr_spill 0") == kernels and "C2 CartPole gravity RandomWalk" in text
    assert json.load(open(os.path.join(d, "manifest.json"))) == manifest
    # the CartPole step kernels keep the 6 wavefronts per SIMD the launch policy is built on (<= 80 VGPRs)
    c1 = next(m for m in manifest.values() if m["what"].startswith("C1 / C5 CartPole"))
    assert c1["kernels"]["nsg_spec_step"]["vgpr_count"] <= 80
    # the NSG_F_LIBM_EXACT units of the same configurations ship too (and the exact C1 step kernel keeps those 6 wavefronts)
    exact = [m for m in manifest.values() if "NSG_F_LIBM_EXACT" in m["what"]]
    assert len(exact) == len(prebuilt.EXACT) + 1
    assert next(m for m in exact if m["what"].startswith("C1 / C5"))["kernels"]["nsg_spec_step"]["vgpr_count"] <= 80


def test_keys_are_stable_and_depend_on_what_they_should(units, tmp_path):
    d, manifest = units
    lib = _lib.load()
    cfg = prebuilt._config("c1", True)
    name_of = lambda n, arch, where: (_lib.check(lib.nsg_spec_prebuild(C.byref(cfg), n, arch, str(where).encode()), "prebuild"),  # noqa: E731
                                      sorted(os.listdir(where)))[1]
    a = tmp_path / "a"; a.mkdir()
    assert name_of(1 << 20, b"gfx950:sramecc+:xnack-", a) == name_of(1 << 19, b"gfx950:sramecc+:xnack-", a)      # same policy: same unit
    assert name_of(1 << 20, b"gfx950:sramecc+:xnack-", a)[0] in manifest                                            # ... the shipped one
    b = tmp_path / "b"; b.mkdir()
    assert name_of(1 << 16, b"gfx950:sramecc+:xnack-", b) != name_of(1 << 20, b"gfx950:sramecc+:xnack-", a)      # in-lane reset range
    c = tmp_path / "c"; c.mkdir()
    assert name_of(1 << 20, b"gfx950", c) != name_of(1 << 20, b"gfx950:sramecc+:xnack-", a)                      # the target id is part of the key


def test_the_in_tree_units_are_what_build_ships():
    """`__graft_entry__.build()` has run in this checkout: the library's own prebuilt directory is populated and current."""
    d = prebuilt.DIR
    assert os.path.isdir(d), "run __graft_entry__.build() (or python -m ns_gym_amd.prebuilt)"
    shipped = json.load(open(os.path.join(d, "manifest.json")))
    assert sorted(f for f in os.listdir(d) if f.endswith(".hsaco")) == sorted(shipped)
