"""The heap step of the parity mode's replay (csrc/cdist_heap_asm.h) is generated: the generator interprets its own block
against a port of libstdc++'s __adjust_heap + __push_heap (what std::partial_sort runs on, reference pt.py:583 through
ATen's topk), and the committed header must be what the generator writes.  No GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_heap_block_is_adjust_heap_and_header_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_heap_asm.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "replacements identical to __adjust_heap" in r.stdout
    assert "committed header matches" in r.stdout
