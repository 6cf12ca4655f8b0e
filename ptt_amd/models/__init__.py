"""Host-side mirror of the reference's module interface for the hot path (ptt/models/...)."""
