"""profiles/probes/ holds the evidence behind negative results: kernel experiments as patches and stand-alone benchmark harnesses.  They are not product code, but they
must not rot silently: every patch still applies to the tree it sits in, every harness still compiles for gfx950 against the current kernel headers.
(A patch that stops applying is either refreshed or deleted together with a note of the last commit that held it — profiles/probes/README.md.)"""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = os.path.join(ROOT, 'profiles', 'probes')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def test_every_probe_patch_still_applies():
    if shutil.which('git') is None:
        pytest.skip('git is not installed here')
    patches = sorted(glob.glob(os.path.join(PROBES, '*.patch')))
    assert patches, 'no patches under profiles/probes'
    bad = []
    for p in patches:
        r = subprocess.run(['git', 'apply', '--check', p], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            bad.append((os.path.basename(p), r.stdout.strip().splitlines()[:2]))
    assert not bad, bad


def test_every_probe_harness_still_compiles():
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc here')
    srcs = sorted(glob.glob(os.path.join(PROBES, '*.hip')))
    assert srcs

    def build(src):
        out = os.path.join('/tmp', 'mcn_probe_' + os.path.basename(src) + '.o')
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O1', '-std=c++17', '-I' + os.path.join(ROOT, 'myconvnet_amd', 'csrc'), '-c', src, '-o', out],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        return os.path.basename(src), r.returncode, r.stdout[-800:]
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(build, srcs))
    bad = [(n, o) for n, rc, o in res if rc != 0]
    assert not bad, bad
