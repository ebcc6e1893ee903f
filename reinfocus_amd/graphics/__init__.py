"""Host-side mirror of reinfocus.graphics for the FastRenderer path."""
