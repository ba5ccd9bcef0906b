"""mtscomp_amd: mtscomp's chunked delta + DEFLATE hot path on AMD Instinct MI355X (gfx950).

Drop-in for the reference's Python API (``compress``, ``decompress``, ``Writer``, ``Reader``) and its
``.cbin`` / ``.ch`` on-disk format; the per-chunk codec runs in hand-written HIP kernels behind the C ABI
declared in ``include/mtscomp_hip.h``.  See DESIGN.md and INTEGRATION.md.
"""
from .api import (  # noqa: F401
    Bunch, CHECK_ATOL, DEFAULT_CONFIG, FORMAT_VERSION, HipCodec, Reader, Writer, add_default_handler, check,
    compress, cumsum_along_axis, decompress, diff_along_axis, get_codec, load_raw_data, read_config,
    set_codec, write_config, __version__)
