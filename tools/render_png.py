"""Render one of the built-in worlds on the GPU and write the tonemapped frame as a PNG (stdlib
only): python tools/render_png.py cornell|cornell_plane|world1|world2|world2_glass W H SPP DEPTH out.png"""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
import raytracer_amd as rt                      # noqa: E402


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    chunk = lambda tag, data: struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def main():
    kind, w, h, spp, depth, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    ctx = rt.Context(0)
    world = bench.make_world(ctx, kind)
    img, st = world.render(w, h, spp, depth)
    packed = ctx.tonemap(img, spp)              # 0x00RRGGBB, main.rs:369-373
    rgb = np.stack([(packed >> 16) & 255, (packed >> 8) & 255, packed & 255], axis=-1).astype(np.uint8).reshape(h, w, 3)
    write_png(out, rgb)
    print(f"{kind} {w}x{h} {spp} spp depth {depth}: {st['rays_traced']} rays in {st['ms_render']:.1f} ms -> {out}")


if __name__ == "__main__":
    main()
