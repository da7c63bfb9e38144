"""SURVEY.md §8(b), threading row: "one handle per GPU; distinct handles may be driven from distinct threads/processes".
One process, two `cl2_renderer` handles on device 0, two host threads driving them at the same time (ctypes releases the
GIL for the duration of a foreign call, so the two `cl2_run_samples` calls really overlap): each handle must end exactly
where it ends when it runs alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(scene, seed, n, barrier=None, out=None, key=None, stage_calls=False):
    from clive2_amd.renderer import Renderer, make_seeds
    r = Renderer(scene, seeds=make_seeds(scene.pixel_width * scene.pixel_height, seed=seed))
    if barrier is not None:
        barrier.wait()                      # both threads enter the library together
    if stage_calls:
        for _ in range(n):
            r.make_light_rays(); r.make_camera_rays(); r.trace_light_rays(); r.trace_camera_rays()
            r.join_paths(); r.finalize_samples(); r.gather_light_image(); r.process_images()
    else:
        r.run_samples(n)
    res = (r.packed_accumulators().copy(), r.get_random_buffer().copy(), r.counters()["rays"],
           r.export_paths(0).tobytes(), r.export_paths(1).tobytes())
    r.close()
    if out is not None:
        out[key] = res
    return res


@pytest.mark.parametrize("scene_name", ["cornell_small", "glass_scene"])
def test_two_handles_two_threads_one_device(scene_name, request):
    scene = request.getfixturevalue(scene_name)
    n = 24
    alone = {k: _render(scene, seed, n) for k, seed in (("a", 101), ("b", 202))}
    assert not np.array_equal(alone["a"][1], alone["b"][1])
    for stage_calls in (False, True):
        got, errors = {}, []
        barrier = threading.Barrier(2)

        def work(key, seed):
            try:
                _render(scene, seed, n if not stage_calls else 3, barrier=barrier, out=got, key=key, stage_calls=stage_calls)
            except Exception as e:              # noqa: BLE001 -- reported below, in the main thread
                errors.append((key, repr(e)))
                try:
                    barrier.abort()
                except Exception:
                    pass

        threads = [threading.Thread(target=work, args=(k, s)) for k, s in (("a", 101), ("b", 202))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        assert all(not t.is_alive() for t in threads)
        if stage_calls:
            ref = {k: _render(scene, seed, 3, stage_calls=True) for k, seed in (("a", 101), ("b", 202))}
        else:
            ref = alone
        for k in ("a", "b"):
            assert np.array_equal(got[k][1], ref[k][1])                     # RNG state
            assert got[k][2] == ref[k][2]                                   # rays traced
            assert got[k][3] == ref[k][3] and got[k][4] == ref[k][4]        # last sample's subpaths, bit for bit
            # accumulators: deterministic sums except the float-atomic light splat (order of additions)
            a, b = got[k][0].reshape(8, -1), ref[k][0].reshape(8, -1)
            assert np.array_equal(a[4:], b[4:])                             # unidirectional estimate + counts: exact
            assert np.allclose(a[:4], b[:4], rtol=2e-5, atol=1e-7)
