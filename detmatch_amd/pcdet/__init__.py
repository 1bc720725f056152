"""PV-RCNN module graph on the MI355X-native operators — host-side mirror of the
parts of thirdparty/Spconv-OpenPCDet/pcdet/models that configs/detmatch select
(NAME='PVRCNN').  Module / parameter names follow the reference so that released
checkpoints load unchanged."""
