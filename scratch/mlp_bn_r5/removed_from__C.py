class FrBnSrc(Structure):       # include/fairrec_hip.h: fr_bn_src
    _fields_ = [("Z", c_void_p), ("fin", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("act", c_int32),
                ("drop_p", c_float), ("drop_seed", c_uint64), ("drop_off", c_uint64)]



    "fr_bnl_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "fr_bnl_fwd": (c_int, [POINTER(FrBnSrc), c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_float, c_float,
                           c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p]),
    "fr_bnl_out": (c_int, [POINTER(FrBnSrc), c_int64, c_int32, c_void_p, c_void_p]),
    "fr_bnl_bwd_top": (c_int, [c_void_p, POINTER(FrBnSrc), c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_void_p, c_void_p]),
    "fr_bnl_bwd": (c_int, [c_void_p, POINTER(FrBnSrc), c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, POINTER(FrBnSrc),
                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
