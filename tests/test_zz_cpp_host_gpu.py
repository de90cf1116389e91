"""The C++ side of the boundary, end to end: examples/render_glb.cpp is a headless render loop written against the
reference's names (include/spatialray_adapter.hpp: FileManager::loadGLTFFile, PackagedScene, RayEngine::load_scene,
trace_pass, FileManager::writeRawFile, ImageHandler::postProcessRaw) in plain C++17 with no HIP headers. The binary
(built by __graft_entry__.build()) renders a .glb on the GPU; the frame it writes must equal the one the Python
mirror renders from the same file with the same seed (same keyed streams; float atomics may reorder additions)."""
import os
import subprocess

import numpy as np
import pytest

from gltf_files import write_test_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_render_loop_matches_the_python_mirror(tmp_path):
    import torch
    from raytracing_amd import engine, gltf, output
    exe = os.path.join(ROOT, "examples", "render_glb")
    assert os.path.exists(exe), "examples/render_glb is missing: run __graft_entry__.build()"
    glb = tmp_path / "scene.glb"
    write_test_scene(str(glb), seed=3, textures=True)
    w, h, subdiv, spp, seed = 64, 36, 2, 4, 77
    raw, png = tmp_path / "out.raw", tmp_path / "out.png"
    run = subprocess.run([exe, str(glb), str(w), str(h), str(subdiv), str(spp), str(seed), str(raw), str(png)],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert "rays" in run.stdout

    desc = gltf.load_gltf_file(str(glb), w, h, subdiv)
    RE = engine.RayEngine(0)
    RE.load_scene(engine.package(desc))
    acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    aux = torch.zeros_like(acc)
    rays = 0
    for p in range(spp):
        rays += int(RE.trace_pass(desc.camera, (0, 0, w, h), p, seed, acc, aux, ray_capacity=1 << 24)["rays_processed"])
    torch.cuda.synchronize()
    assert f"{rays} rays" in run.stdout                       # the same ray tree
    got, rx, ry = output.read_raw(str(raw))
    assert (rx, ry) == (w, h)
    want = acc.cpu().numpy()[:, :3].reshape(h, w, 3)
    assert np.allclose(np.asarray(got).reshape(h, w, 3), want, rtol=1e-4, atol=1e-6)
    assert png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n" and png.stat().st_size > w * h * 4
    # a missing file is an error message and a non-zero exit, not a crash
    bad = subprocess.run([exe, str(tmp_path / "nope.glb"), "8", "8", "1", "1", "1", str(raw)], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "cannot open" in bad.stderr
    RE.close()
