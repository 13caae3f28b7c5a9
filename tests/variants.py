"""Kernel instantiations the parity tests run every case through (environment switches read by every batch call)."""
# the instantiations of the heavy-item lane kernel (engine.hip): two / three waves per SIMD over global regions (k_lift_lanes_g, _w3), and
# the streaming kernel -- a team of waves per 64 item slots, the stages chained through LDS rings (k_lift_stream; lane_stream.hpp)
HEAVY_VARIANTS = {"g": {"PLO_LANE_STREAM": "0", "PLO_LANE_G_W3": "0"}, "g_w3": {"PLO_LANE_STREAM": "0", "PLO_LANE_G_W3": "1"},
                  "stream": {"PLO_LANE_STREAM": "1", "PLO_LANE_G_W3": "0"}}
