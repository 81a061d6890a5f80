"""Import-path aliases so the reference experiment JSONs (dotted class paths) resolve to sound_bubble_amd."""
