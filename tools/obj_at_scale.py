"""Writes an n-triangle soup as a Wavefront OBJ file (one object, triangles that share no vertices),
loads it back through the library's OBJ ingest (rt_world_create_cornell_box: Mesh::from_file +
the flattening, host only) and prints the times:  python tools/obj_at_scale.py <n_tris> [dir]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt                      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    out_dir = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
    path = os.path.join(out_dir, f"soup_{n}.obj")
    rng = np.random.default_rng(7)
    t0 = time.time()
    mtl = os.path.join(out_dir, f"soup_{n}.mtl")      # (create_cornell_box takes its materials from the file's MTL)
    with open(mtl, "w") as f:
        f.write("newmtl white\nNs 0.0\nKa 1.0 1.0 1.0\nKd 0.8 0.8 0.8\nKs 0.0 0.0 0.0\nKe 0.0 0.0 0.0\nNi 1.45\nd 1.0\nillum 2\n")
    with open(path, "w") as f:
        f.write(f"mtllib {os.path.basename(mtl)}\no soup\nusemtl white\n")
        chunk = 500_000
        for base in range(0, n, chunk):
            m = min(chunk, n - base)
            c = rng.uniform(-50.0, 50.0, (m, 1, 3))
            p = (c + rng.uniform(-0.4, 0.4, (m, 3, 3))).astype(np.float32).reshape(-1, 3)
            f.write("".join(f"v {x:.6f} {y:.6f} {z:.6f}\n" for x, y, z in p))
            f.write("vn 0.000000 0.000000 1.000000\n" * (3 * m))      # (the reference reads a normal per vertex)
            idx = np.arange(base * 3 + 1, (base + m) * 3 + 1).reshape(-1, 3)
            f.write("".join(f"f {a}//{a} {b}//{b} {c_}//{c_}\n" for a, b, c_ in idx))
            print(f"  written {base + m} triangles", flush=True)
    t_write = time.time() - t0
    size = os.path.getsize(path)
    t0 = time.time()
    tris, models, faces = rt.obj_count(path)
    t_count = time.time() - t0
    t0 = time.time()
    w = rt.World.cornell_box(None, path)
    t_world = time.time() - t0
    print(f"{n} triangles: file {size / 1e6:.0f} MB written in {t_write:.1f} s; counted ({tris} triangles, {models} models) in "
          f"{t_count:.2f} s; parsed and flattened ({w.scene.num_entries} entries) in {t_world:.2f} s "
          f"on {len(os.sched_getaffinity(0))} host threads")
    w.close()
    os.remove(path)
    os.remove(mtl)


if __name__ == "__main__":
    main()
