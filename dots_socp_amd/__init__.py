"""dots_socp_amd: MI355X-native ALM hot path of DOTs-SOCP (dynamic optimal transport on
triangulated surfaces as a linear second-order-cone program).

Host side (this package, Python like the reference) mirrors the reference's solver
plug-in API -- ``solver_socp`` / ``solver_raw`` / ``solver`` with the signature of
``dot_surface_socp/socp/solver_socp.py:25-41`` and ``socp/__init__.py:6-11`` -- and
drives hand-written HIP kernels for gfx950 through the C ABI declared in
``include/dots_socp_hip.h`` (``csrc/`` builds ``libdotsocp_hip.so``).
There is no CPU fallback: without the HIP library every solver call raises.
"""
__version__ = "0.3.0"
