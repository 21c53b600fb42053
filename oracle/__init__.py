"""CPU oracle of the RetinaUNet hot path -- TEST INFRASTRUCTURE ONLY (see oracle/README.md)."""
