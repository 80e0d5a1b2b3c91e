"""GPU: error behaviour of the C ABI -- every misuse returns an error code + message (PmxError in the binding), nothing
crashes, and the context stays usable afterwards."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture()
def small(native):
    eng = native.Engine(0, max_batch=2, max_h=64, max_w=96)
    yield eng
    eng.close()


def test_missing_weights_is_an_error_not_garbage(native, small):
    img = np.zeros((1, 64, 96, 3), np.uint8)
    assert small.weights_missing() == 92
    with pytest.raises(native.PmxError) as e:
        small.forward_u8(img)
    assert 'weights' in str(e.value)


def test_capacity_and_shape_violations(native, small):
    small.set_weights(pkg('weights').synthetic_weights(0))
    with pytest.raises(native.PmxError):
        small.forward_u8(np.zeros((3, 64, 96, 3), np.uint8))            # batch > max_batch
    with pytest.raises(native.PmxError):
        small.forward_u8(np.zeros((1, 128, 96, 3), np.uint8))           # larger than the context
    with pytest.raises(native.PmxError):
        small.forward_u8(np.zeros((1, 60, 96, 3), np.uint8))            # not a multiple of 8
    # the context still works
    small.forward_u8(np.zeros((2, 64, 96, 3), np.uint8))
    paf, heat = small.get_maps()
    assert paf.shape == (2, 38, 8, 12) and np.isfinite(paf).all() and np.isfinite(heat).all()


def test_wrong_layer_shape_or_name(native, small):
    w = np.zeros((64, 3, 3, 3), np.float32)
    b = np.zeros(64, np.float32)
    with pytest.raises(native.PmxError):
        small.set_layer('no_such_layer', w, b)
    with pytest.raises(native.PmxError):
        small.set_layer('conv1_1', np.zeros((64, 4, 3, 3), np.float32), b)      # cin mismatch
    with pytest.raises(native.PmxError):
        small.set_layer('conv1_1', np.zeros((32, 3, 3, 3), np.float32), np.zeros(32, np.float32))   # cout mismatch
    small.set_layer('conv1_1', w, b)


def test_state_errors(native, small):
    with pytest.raises(native.PmxError):
        small.postprocess(56, 80, img_len=80)                            # no maps yet
    rc = small.lib.pmx_precise_finish(small._ctx)
    assert rc != 0 and b'nothing accumulated' in small.lib.pmx_last_error()
    assert small.lib.pmx_set_option(small._ctx, b'no_such_option', 1) != 0
    # capacities: people <= subsets, bounded; a refused call leaves the context as it was
    before = small.capacities()
    assert small.lib.pmx_set_capacities(small._ctx, 0, 2, 8, 0) == 1 and b'must not exceed' in small.lib.pmx_last_error()
    assert small.lib.pmx_set_capacities(small._ctx, 1 << 21, 0, 0, 0) == 1
    assert small.capacities() == before
    # the device pointer of the records is only handed out once a post-process has produced them
    import ctypes as C
    p, nb = C.c_void_p(), C.c_size_t()
    fresh = native.Engine(0, max_batch=1, max_h=64, max_w=64)
    assert fresh.lib.pmx_results_device_ptr(fresh._ctx, C.byref(p), C.byref(nb)) == 6       # PMX_ERR_STATE
    fresh.close()


def test_null_arguments_do_not_crash(native):
    lib = native.load()
    assert lib.pmx_forward_u8(None, None, 1, 64, 64, 0) != 0
    assert lib.pmx_get_results(None, 1, None, 0) != 0
    assert lib.pmx_results_layout(None, None, None) != 0
    assert lib.pmx_set_capacities(None, 0, 0, 0, 0) != 0
    assert lib.pmx_create(None, 0, 1, 64, 64) != 0
    assert lib.pmx_create_net(None, b'posenet', 0, 1, 64, 64) != 0
    lib.pmx_destroy(None)                                                # no-op
