"""shallow-ntc on MI355X: the eval-time hot path of mandt-lab/shallow-ntc as hand-written HIP.

Layout mirrors the reference so call sites read the same:
    common.transforms.class_builder   <- reference common/transforms.py:380-393
    mshyper.models.Model              <- reference mshyper/models.py
    factorized.models.Model           <- reference factorized/models.py
The arithmetic lives in lib/libsntc_hip.so (csrc/, C ABI in include/sntc.h); this package is the
thin driver.  Import name: ``shallow_ntc_amd`` (see __graft_entry__.load_package()).
"""
__version__ = "0.1.0"
