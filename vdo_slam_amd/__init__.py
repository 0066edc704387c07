"""vdo_slam_amd — MI355X-native (HIP/gfx950) hot path of VDO-SLAM behind a C-ABI.

Python is only the test/bench host: ``_capi`` binds ``libvdo_hip.so`` with ctypes,
``synth`` generates KITTI-shaped synthetic inputs.  The C++ mirror of the reference's
``System / Tracking / Optimizer / ORBextractor`` classes lives in ``vdo_slam_amd/host``.
"""
__version__ = "0.1.0"
