"""The GLB importer under AddressSanitizer and UndefinedBehaviorSanitizer on damaged files.

tests/hostcheck/fuzz_glb is wr_gltf.cpp + wr_jpeg.cpp (host-only C++: JSON, accessors, PNG
through zlib, baseline and progressive JPEG; the counterpart of gltf::import_slice and the image
crate at src/core/resources.rs:28) built on their own with g++ -fsanitize=address,undefined and a
driver that mutates a seed file (bit flips, overwritten length fields, truncations, spliced
runs, edits of the JSON text) and walks whatever still parses through every accessor of the C
ABI.  Any out-of-bounds access, overflow or exception through the ABI aborts the program.
CPU only."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = os.path.join(ROOT, "tests", "hostcheck", "fuzz_glb")


def _run(seed_file, rounds, rng):
    if not os.path.exists(FUZZ):
        pytest.fail("tests/hostcheck/fuzz_glb is not built: make -C tests/hostcheck")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([FUZZ, str(seed_file), str(rounds), str(rng)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
    mutations, parsed = (int(x) for x in (p.stdout.split()[0], p.stdout.split()[2]))
    assert mutations == rounds and 0 < parsed < rounds  # some damage is refused, some is survivable
    return parsed


@pytest.mark.parametrize("rng", [3, 77])
def test_damaged_small_files(tmp_path, rng):
    """Every feature of the synthetic file: interleaved and normalised accessors, 8/16/32-bit
    indices, RGBA / RGB / palette PNG, unicode and escapes in the JSON."""
    from test_abi_and_host import _synthetic_glb
    seed = tmp_path / "seed.glb"
    seed.write_bytes(_synthetic_glb(np.random.default_rng(7), True)[0])
    _run(seed, 12000, rng)


def test_damaged_file_with_jpeg_images(tmp_path):
    """A small file of Sponza's structure: baseline, 4:4:4 and progressive JPEG next to PNG."""
    from sponza_like import sponza_like_glb
    glb, info = sponza_like_glb(detail=0.03, tex_size=16, primitives=30)
    assert {"jpeg", "jpeg_444", "jpeg_progressive"} <= set(info["encodings"])
    seed = tmp_path / "seed.glb"
    seed.write_bytes(glb)
    _run(seed, 2500, 5)
