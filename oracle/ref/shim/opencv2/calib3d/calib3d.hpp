// TEST INFRASTRUCTURE - stand-in for the OpenCV 3.4 header of the same name (OpenCV is not installed in this image):
// lets the reference's own sources compile where they lie.  Everything is in minicv_ref.hpp.
#include "minicv_ref.hpp"
