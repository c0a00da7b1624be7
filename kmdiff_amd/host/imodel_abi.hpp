// imodel_abi.hpp -- the reference's model-plugin surface, declared from its published contract
// so that a plugin built here is loadable by kmdiff's plugin_manager (model_manager.hpp:33-94):
//
//   template<size_t MAX_C> class IModel      include/kmdiff/imodel.hpp:23-72
//       virtual ~IModel(); virtual void configure(const std::string&) = 0;
//       virtual model_ret_t process(const Range<count_type>&, const Range<count_type>&) = 0;
//   Range<T>  = { std::vector<T>& data; size_t start; size_t size; }   include/kmdiff/range.hpp:7-35
//   model_ret_t = std::tuple<double, Significance, double, double>      imodel.hpp:17
//   Significance { CONTROL, CASE, NO }                                   kmer.hpp:33-38
//   count_type = uint8_t | uint16_t | uint32_t for MAX_C <= 255 | <= 65535 | larger (km::selectC)
//   exported: extern "C" std::string plugin_name(); IModel<maxcN>* createN();   plugins/ex_model.cpp:29-32
//
// The vtable order (destructor, configure, process) and the member order of Range are the ABI.
#pragma once
#include <cstddef>
#include <cstdint>
#include <limits>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

namespace kmdiff {

enum class Significance { CONTROL, CASE, NO };

template <typename T>
class Range
{
  std::vector<T>& m_data;
  size_t m_start;
  size_t m_size;
 public:
  Range(std::vector<T>& data, size_t start, size_t size) : m_data(data), m_start(start), m_size(size) {}
  auto begin() const { return m_data.begin() + m_start; }
  auto end() const { return m_data.begin() + m_start + m_size; }
  size_t size() const { return m_size; }
  const T& operator[](size_t i) const { return m_data[m_start + i]; }
};

using model_ret_t = std::tuple<double, Significance, double, double>;

constexpr size_t maxc8 = std::numeric_limits<uint8_t>::max();
constexpr size_t maxc16 = std::numeric_limits<uint16_t>::max();
constexpr size_t maxc32 = std::numeric_limits<uint32_t>::max();

template <size_t MAX_C>
struct select_count
{
  using type = std::conditional_t<(MAX_C <= maxc8), uint8_t, std::conditional_t<(MAX_C <= maxc16), uint16_t, uint32_t>>;
};

template <size_t MAX_C>
class IModel
{
 public:
  using count_type = typename select_count<MAX_C>::type;
  using range_type = Range<count_type>;
  IModel() = default;
  virtual ~IModel() {}
  virtual void configure(const std::string& config) = 0;
  virtual model_ret_t process(const range_type& controls, const range_type& cases) = 0;
};

} // namespace kmdiff
