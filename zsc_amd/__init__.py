"""zsc_amd -- MI355X-native DEFLATE hot path behind zsc's own API.

Host-side mirror (Python, ctypes) of the C ABI exported by ``libzsc_hip.so``:

* :func:`compress`, :func:`compress2`, :func:`compress_gzip`, :func:`uncompress` ... --
  the reference's one-shot functions (``include/zsc/zsc_pub.h``), same argument
  meaning and ``ZlibReturn`` codes;
* :func:`compress_batch` -- many independent buffers per call;
* :class:`DeflatePlan` -- device-resident batches (inputs and outputs stay in HBM).

There is no CPU codec here: if the HIP library is missing, import fails loudly.
"""
from .api import (  # noqa: F401
    Z_OK, Z_STREAM_END, Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR,
    Z_DEFAULT_STRATEGY, Z_FILTERED, Z_FIXED, GZIP_CODE, DEF_WBITS, DEF_MEM_LEVEL,
    lib, lib_path, build_library, device_info,
    compress_get_min_work_buf_size, compress_get_max_output_size, compress_get_max_output_size2,
    uncompress_get_min_work_buf_size,
    compress, compress2, compress_gzip, uncompress, uncompress2, uncompress_gzip,
    compress_batch, compress_sections_batch, compress_sections_device, uncompress_batch, DeflatePlan, InflatePlan,
    GzHeader, gz_header_for_writing, gz_header_for_reading, gz_header_fields,
)
