"""Work partition of one frame over the ranks of a multi-GPU job.

Every (pixel, sample) of the reference's loop (main.rs:358-367) is independent, so a frame is
split two ways at once: by image tile (interleaved bands of BAND_ROWS rows, the reference's
64-pixel block height, main.rs:22) and by sample range.  Rank r of N renders band group
r % n_band_groups and sample part r // n_band_groups; every rank accumulates into its own
full-frame buffer and one sum-reduce to rank 0 combines them.
"""

BAND_ROWS = 64


def split_factors(world_size, spp):
    """(band groups, sample parts) with band_groups * sample_parts == world_size."""
    if world_size < 1:
        raise ValueError("world_size must be positive")
    bands = 2 if (world_size >= 4 and world_size % 2 == 0) else 1
    parts = world_size // bands
    if parts > spp:                      # fewer samples than parts: split by bands only
        bands, parts = world_size, 1
    return bands, parts


def sample_range(spp, parts, part):
    """[begin, end) of sample part `part` of `parts`: as even as possible, in order."""
    base, extra = divmod(spp, parts)
    begin = part * base + min(part, extra)
    return begin, begin + base + (1 if part < extra else 0)


def band_rects(width, height, groups, group):
    """Pixel rectangles (x0, y0, x1, y1) of band group `group`: bands group, group+groups, ..."""
    rects = []
    n_bands = (height + BAND_ROWS - 1) // BAND_ROWS
    for b in range(group, n_bands, groups):
        rects.append((0, b * BAND_ROWS, width, min(height, (b + 1) * BAND_ROWS)))
    return rects


def partition(width, height, spp, world_size, rank):
    """-> (bands, (sample_begin, sample_end)) for `rank`, where bands = (band_rows, band_count,
    band_index) is the interleaved band group of the full frame this rank renders (the renderer
    takes it in one call, rt_render_params.band_*); band_rects() lists the same pixels as
    rectangles."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    groups, parts = split_factors(world_size, spp)
    group, part = rank % groups, rank // groups
    return (BAND_ROWS, groups, group), sample_range(spp, parts, part)
