from neddf_amd.loss import BaseLoss, ColorLoss, FieldsConstraintLoss, MaskBCELoss, MaskMSELoss  # noqa: F401
