"""ISP filter stack (host-side mirror of the reference's `isp` package; pixels run in HIP)."""
