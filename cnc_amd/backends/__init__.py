"""Host-side mirrors of the reference's three compiled extensions, on top of libcnc_hip.so.

    gridencoder_backend  <->  `_gridencoder`     (gridencoder/src/bindings.cpp:5-9)
    pack_and_align       <->  `pack_and_align`   (my_cuda_backen/aligner.cpp:73-78)
    nerfacc_cuda         <->  `nerfacc.csrc`     (nerfacc/cuda/csrc/nerfacc.cpp:100-129)

`cnc_amd.install_dropins()` registers them in sys.modules under the reference's import names.
"""
