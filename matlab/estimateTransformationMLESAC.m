function [tform, inlierIdx, isFound] = estimateTransformationMLESAC(points1, points2, transformationType, input)
    %ESTIMATETRANSFORMATIONMLESAC Shadows PP/imageMatching/estimateTransformationMLESAC.m (all five transformationTypes).
    %   The minimal-sample draws are generated here with randperm exactly as the reference does (:160) and handed to
    %   the device as an explicit input (a 4-row column per loop iteration, the first sampleSize rows are the sample);
    %   fitting, truncated-loss scoring, the adaptive stop and the refit on the inliers run in aps_mex.
    if nargin < 4, input = struct(); end
    switch lower(transformationType)  % sampleSize (:83-92)
        case 'projective', k = 4;
        case 'affine', k = 3;
        case {'similarity', 'rigid'}, k = 2;
        case 'translation', k = 1;
        otherwise, error('Unknown transform type');
    end
    if size(points1, 1) ~= size(points2, 1)
        error('estimateTransformationMLESAC:PointCountMismatch', 'points1 and points2 must have the same number of rows.');
    end
    M = size(points1, 1);
    if M < k
        tform = []; inlierIdx = false(M, 1); isFound = false; return;
    end
    if isfield(input, 'maxIter'), S = input.maxIter + 64; else, S = 1064; end
    sampleIdx = ones(4, S, 'uint32');
    for s = 1:S, sampleIdx(1:k, s) = uint32(randperm(M, k)); end
    [tform, inlierIdx, isFound] = aps_mex('mlesac_homography', double(points1), double(points2), input, sampleIdx, lower(transformationType));
    if ~isFound, tform = []; end
end
