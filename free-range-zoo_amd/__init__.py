"""MI355X-native batched environment stepping for free-range-zoo's wildfire / rideshare / cybersecurity domains.

Python mirrors the reference's interface (``envs.<domain>_v0.parallel_env``, ``Configuration`` dataclasses,
``reset/step/observe``); all arithmetic runs in hand-written HIP kernels behind the C-ABI of ``include/frz.h``.
Import as ``free_range_zoo_amd``.
"""
__version__ = '0.1.0'
