"""The benchmark's pass-A kernels must not spill: a change that pushes them over their 128 vector registers shows up
as 3 ms per launch on the GPU (the overflow path's bin-width constants, the ragged / mask code compiled into
every variant) and as nothing at all in the parity tests.  hipcc cross-compiles without a GPU, so the compiler's own
kernel-resource remarks are checked here for the h = 4 translation unit (tools/kres.py prints the whole table)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "btl_bloomfilter_amd", "csrc", "part_hash_inst.hip")


@pytest.fixture(scope="module")
def remarks(tmp_path_factory):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    assert os.path.exists(hipcc), "hipcc is part of the image: the build check needs it too"
    obj = tmp_path_factory.mktemp("kres") / "part_hash_h4.o"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                        "-Rpass-analysis=kernel-resource-usage", "-DBTLBF_PART_H=4", "-c", "-o", str(obj), SRC],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("btlbf::", "")
            out[cur] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                out[cur][key] = int(m.group(1))
    return out


# <H, POW2, SPACED, QUERY, WINDOW, AUX>
@pytest.mark.parametrize("kernel", ["part_hash_ov_kernel<4, true, false, false, false, false>",   # C2 insert
                                    "part_hash_ov_kernel<4, true, false, true, false, false>",    # C2 query
                                    "part_hash_ov_kernel<4, true, false, false, false, true>"])   # ragged insert
def test_headline_pass_a_kernels_do_not_spill(remarks, kernel):
    assert kernel in remarks, sorted(remarks)[:5]
    res = remarks[kernel]
    assert res["vgpr"] <= 128
    assert res["scratch"] == 0, "%s spills %d bytes per lane: python tools/kres.py part_hash_inst.hip -DBTLBF_PART_H=4" % (kernel, res["scratch"])
