"""repmode_amd -- MI355X-native (gfx950) MoDE-block hot path of RepMode.

Drop-in surface: ``repmode_amd.nn_modules.RepMode.Net(opts)`` / ``net(signal, task)`` with the
reference's 309-key ``state_dict``; native boundary: ``include/repmode_hip.h`` (C ABI of
``librepmode_hip.so``, hand-written HIP kernels).  See DESIGN.md and INTEGRATION.md.
"""
__version__ = '0.1.0'
