"""Static check of the compiled device code (no GPU needed: hipcc cross-compiles): no instruction writes a data register of a
128-bit buffer store in the two slots after the store.  On gfx950 such a write reaches memory in place of the stored value for
the lanes the store reads last; hipcc pads the hazard only for stores without a register soffset, and every bulk store of the
kernels has one (csrc/cg_kernels.h, buf_store).  Found with the fp32 kernels at 3 rows in flight: tests/test_gpu_mixed.py."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_write_to_the_data_registers_right_after_a_128_bit_buffer_store(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_store_hazard_check as chk
    os.environ["PATH"] = os.environ.get("PATH", "") + ":/opt/rocm/bin"
    dump = str(tmp_path / "dev.s")
    chk.compile_to_asm(dump)
    text = open(dump).read()
    assert text.count("buffer_store_dwordx4") > 50                      # the check looked at the real kernels
    bad = chk.findings(dump)
    assert not bad, "\n".join(f"{k}: {st} -> {wr} after {n} wait state(s)" for k, st, wr, n in bad)

