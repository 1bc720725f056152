// ORACLE — TEST INFRASTRUCTURE ONLY.
// pybind stub around the reference's own rotated-BEV-IoU CPU implementation
// (thirdparty/Spconv-OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232 boxes_iou_bev_cpu),
// compiled where it lies by oracle/build_ref.py.  The reference's module definition
// (iou3d_nms_api.cpp) also binds the GPU entry points, which cannot link here, so only the
// CPU function is bound.
#include <torch/extension.h>
#include "iou3d_cpu.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("boxes_iou_bev_cpu", &boxes_iou_bev_cpu);
}
