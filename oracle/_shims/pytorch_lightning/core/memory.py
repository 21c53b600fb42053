class ModelSummary:
    pass
