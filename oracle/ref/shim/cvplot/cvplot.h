// TEST INFRASTRUCTURE - stand-in for the reference's vendored include/cvplot/cvplot.h (a plotting helper on top of OpenCV's highgui):
// Tracking::PlotMetricError has to compile; nothing is drawn.
#ifndef VDO_REF_CVPLOT_STUB_H_
#define VDO_REF_CVPLOT_STUB_H_
#include <string>
#include <utility>
#include <vector>
namespace cvplot {
enum Type { Line, DotLine, Dots, FillLine, RangeLine, Histogram, Vistogram, Horizontal, Vertical, Range, Circle };
enum Color { Black, Red, Green, Blue, Cyan, Purple, Pink, Orange, Gray, Yellow };
class Series {
 public:
  Series& type(Type) { return *this; }
  Series& color(Color) { return *this; }
  template <typename T> Series& set(const T&) { return *this; }
  template <typename T> Series& setValue(const T&) { return *this; }
  template <typename T> Series& add(const T&) { return *this; }
  template <typename T> Series& addValue(const T&) { return *this; }
  template <typename A, typename B> Series& addValue(const A&, const B&) { return *this; }
  Series& dynamicColor(bool) { return *this; }
  Series& legend(bool) { return *this; }
};
class Figure {
 public:
  Series& series(const std::string&) { return s_; }
  Figure& origin(bool, bool) { return *this; }
  Figure& square(bool) { return *this; }
  Figure& border(int) { return *this; }
  Figure& alpha(float) { return *this; }
  Figure& gridSize(int) { return *this; }
  void show(bool = true) {}
  void clear() {}
 private:
  Series s_;
};
inline Figure& figure(const std::string&) { static Figure f; return f; }
inline void setWindowTitle(const std::string&, const std::string&) {}
inline void moveWindow(const std::string&, int, int) {}
inline void resizeWindow(const std::string&, int, int) {}
}  // namespace cvplot
#endif
