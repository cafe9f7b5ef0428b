"""libmcn_cpu.so — the C-ABI of include/mcn.h as plain C++ / OpenMP loops (myconvnet_amd/csrc_cpu/mcn_cpu.cpp; SURVEY.md section 7 step 2,
section 8b) — lets the HOST code of the product (graph recording, fusion lowering, launch lists, optimizer, training loop, data-parallel
exchange) execute whole training steps in the GPU-less build container.  The binding loads it only when MCN_LIB_PATH names it (there is
no fallback to it: tests/test_abi.py::test_missing_library_fails_loudly), so every case runs in a fresh process with that variable set
(tests/cpu_lib_cases.py) and is compared with the oracle there."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, 'myconvnet_amd', 'libmcn_cpu.so')


@pytest.fixture(scope='module')
def cpu_lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_mcn_build', os.path.join(ROOT, 'myconvnet_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        return mod.build_cpu()
    except Exception as e:                                  # noqa: BLE001  no g++ / libgomp on this host: the library is test infrastructure
        pytest.skip('libmcn_cpu.so cannot be built here: {}'.format(str(e).splitlines()[0]))


def test_cpu_library_exports_every_symbol_of_the_header(cpu_lib):
    from test_abi import declared_symbols
    lib = ctypes.CDLL(cpu_lib)
    for s in declared_symbols():
        assert hasattr(lib, s), 'libmcn_cpu.so does not export {}'.format(s)
    lib.mcn_build_id.restype = ctypes.c_char_p
    assert lib.mcn_build_id() == b'cpu' and lib.mcn_version() == 100


def test_the_product_never_loads_the_cpu_library_by_itself(cpu_lib, tmp_path):
    """without MCN_LIB_PATH the binding is libmcn_hip.so (this process), and a missing libmcn_hip.so is an ImportError even with libmcn_cpu.so
    sitting next to where it should be — there is no fallback"""
    from myconvnet_amd import _ffi
    if not os.environ.get('MCN_LIB_PATH'):
        assert _ffi.LIB_PATH.endswith('libmcn_hip.so') and not _ffi.IS_CPU_LIB
    import shutil
    shutil.copy(cpu_lib, str(tmp_path / 'libmcn_cpu.so'))
    with pytest.raises(ImportError, match='no CPU fallback'):
        _ffi.load(str(tmp_path / 'libmcn_hip.so'))
    # and a model on a CPU device does not compile against the HIP library
    if not _ffi.IS_CPU_LIB:
        import myconvnet_amd as M
        model = M.ResNet18([32, 32, 3], 10, batch_size=4, width_div=8, num_gpus=1, device='cpu')
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            model.compile()


@pytest.mark.parametrize('case', ['resnet50', 'resnet18', 'resnet18_bf16', 'resnet50_fp16', 'resnet18_decay_clip', 'resnet18_frozen_clip', 'resnet18_l1_focal', 'resnet18_l1_clip', 'dw_mult_bias', 'train_loop', 'efficientnet', 'deeplab', 'deeplab_ls', 'deeplab_level', 'dist2', 'dist2_frozen'])
def test_host_code_executes_on_the_cpu_library(case, cpu_lib):
    env = dict(os.environ, MCN_LIB_PATH=cpu_lib, OMP_NUM_THREADS='4')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'cpu_lib_cases.py'), case], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    assert ' ok' in r.stdout.splitlines()[-1], r.stdout
