"""scripts/heapguard.c — the malloc interposer that found the HIP runtime's write after free at stream destruction
(profiles/r06_segv_hunt.txt) — still does what it is for: it reports a write into a freed block with the block's size, the offset
and who freed it, and the library's host side (tapes, their generated code) runs clean under it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

UAF = r"""
#include <stdlib.h>
#include <stdio.h>
struct thing { void* vptr; long count; char rest[80]; };
int main(void) {
    struct thing* t = malloc(sizeof *t);
    t->count = 3;
    free(t);
    t->count--;                                  /* a stale reference-count release */
    for (int i = 0; i < 100; ++i) free(malloc(64));
    puts("done");
    return 0;
}
"""


@pytest.fixture(scope="module")
def guard(tmp_path_factory):
    d = tmp_path_factory.mktemp("heapguard")
    so = str(d / "heapguard.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "scripts", "heapguard.c"), "-ldl"])
    return d, so


def test_a_write_after_free_is_reported_with_its_offset_and_who_freed_the_block(guard):
    d, so = guard
    (d / "uaf.c").write_text(UAF)
    subprocess.check_call(["gcc", "-O0", "-o", str(d / "uaf"), str(d / "uaf.c")])
    r = subprocess.run([str(d / "uaf")], env=dict(os.environ, LD_PRELOAD=so), capture_output=True, text=True)
    assert r.returncode == 0 and "done" in r.stdout
    assert "HEAPGUARD write after free" in r.stderr and "first changed byte at offset 8" in r.stderr and "uaf+0x" in r.stderr, r.stderr
    assert "1 reports" in r.stderr


def test_tapes_and_their_generated_code_under_the_interposer(guard):
    d, so = guard
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import mpr_amd as m\n"
            "for name in ('bear', 'architecture', 'prospero', 'involute_gear_3d'):\n"
            "    for _ in range(3): t = m.Tape(m.model(name)); del t\n"
            "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LD_PRELOAD=so), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    assert "write after free" not in r.stderr, r.stderr[-2000:]
    assert "0 reports" in r.stderr
